// conv_x3.hip -- fp32 3x3 convolution (forward and backward-data) on the bf16 matrix pipe, gfx950.
//
// Same contract as conv_igemm_kernel<float, CONV_3X3, ...> (conv_igemm.hip): fp32 NHWC activations, fp32 packed
// weights [tap][Nout][Cin], fp32 accumulation, fp32 results -- replaces nn.Conv2d(3x3, p=1) + max(0.2x, x) of
// models/arch/Unet.py:11-46,102-104 and its autograd backward-data.  Only the way a product a*w is formed differs:
//
//   every fp32 operand is cut EXACTLY into three bf16 pieces while it is staged into LDS,
//        a = a1 + a2 + a3,   a1 = top 16 bits of a,  a2 = top 16 bits of (a - a1),  a3 = a - a1 - a2
//   (8 + 8 + 8 significant bits; both subtractions are exact in fp32), and
//        a*w ~= a1 w1 + a1 w2 + a2 w1 + a1 w3 + a2 w2 + a3 w1
//   is accumulated in fp32 by six v_mfma_f32_32x32x16_bf16 (each bf16 x bf16 product is exact in fp32).  The dropped
//   terms a2 w3 + a3 w2 + a3 w3 are below 2^-23 |a w|: the error of one product is smaller than the rounding of
//   the fp32 product itself, and the sum is accumulated in fp32 exactly as on the fp32 MFMA -- measured error against
//   an fp64 convolution is the same as (slightly below) the v_mfma_f32_32x32x2_f32 kernel's (tests/test_unet_gpu.py).
//
// Why: six bf16 MFMAs cover 16 k-values in 6 x 32 = 192 matrix-pipe cycles, eight fp32 MFMAs need 8 x 64 = 512 for
// the same 16 k-values -> 2.67x the fp32 peak (2.5 PFLOP/s / 6 = 417 TFLOP/s "fp32-equivalent" against 157 TFLOP/s),
// and each staged fragment feeds 2-3 MFMAs instead of 1, so the LDS read rate per MFMA is half that of the plain
// bf16 kernel.
//
// Tiling: 256 threads = 4 waves, output tile 4*RPW rows x 32 pixels x BN channels, D[channel][pixel] roles as in
// conv_igemm.hip.  K chunk = 16 input channels.  LDS row = [3 pieces][16 bf16] = 96 B + 16 B pad (28-word stride:
// 28/4 is odd, so 16 consecutive rows hit 16 distinct 4-bank groups -> conflict-free ds_read_b128).  The halo tile
// ((TH+2) x 34 pixels) is staged once per chunk; the weight slab is staged per kernel ROW (3 taps x BN x 16), so a
// workgroup needs 38 KB + 21 KB of LDS and two of them share a CU: one stages/synchronises while the other feeds the
// matrix pipe.  Work items (tile, chunk, kernel row) stream through a register-staged pipeline like the fp32 kernel's.
#include <stdlib.h>
#include "conv.h"

#include "conv_x3_dev.h"

namespace {

// BSLAB (round 6): the weights arrive PRE-SPLIT in conv_x3d_kernel's slab layout (x3_store, unet_misc.hip: the exact LDS image of a stage's slab, rows of
// 3 pieces x 16 bf16 + pad) and a stage's slab is copied global -> registers -> LDS as it is (three 16-byte units per thread): no weight cut per
// stage and tile (a fifth of this kernel's VALU work and a third of its ds_write instructions); the register staging and the two 4-wave workgroups
// per CU stay (the LDS-DMA variants of this layer shape, conv_x3d_kernel<32, ...>, need a second slab buffer and lose the second workgroup).
template <int BN, int RPW, bool DB, bool BFIRST = false, bool BSLAB = false, bool STREAM = false>
__global__ __launch_bounds__(256, 2) void conv_x3_kernel(const ConvArgs a) {
    constexpr int CK = 16;
    constexpr int TH = 4 * RPW, NT = BN / 32, A_PIX = (TH + 2) * (TW + 2);
    constexpr int A_WORDS = A_PIX * PX, B_ROWS = 3 * BN;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* ldsA = lds;
    float* ldsB = lds + A_WORDS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = lane & 31, hi = lane >> 5;
    const int NB = a.Nout / BN;
    const int Cin = a.C0 + a.C1;
    const int total_tiles = a.tiles_x * a.tiles_y * a.N * NB;
    const int Cs0 = a.C0;

    constexpr int A_UNITS = A_PIX * 4, B_UNITS = BSLAB ? B_ROWS * 7 : B_ROWS * 4;      // BSLAB: 16-byte units of the slab image (112-byte rows)
    constexpr int A_IT = (A_UNITS + 255) / 256, B_IT = (B_UNITS + 255) / 256;
    constexpr int SLAB_BYTES = (B_ROWS * PX * 4 + 1023) / 1024 * 1024;                      // x3_slab_stride(BN)
    float4 ra[A_IT], rb[B_IT];
    constexpr unsigned OOB = 0xFFFFFFF0u;
    unsigned a_voff[A_IT], b_voff[B_IT];
    int l_img = 0;
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc((void*)a.wp, 0, BSLAB ? (int)((size_t)3 * (Cin / CK) * NB * SLAB_BYTES) : (int)((size_t)9 * a.Nout * Cin * 4), 0x00020000);
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
        const int u = tid + it * 256;
        if constexpr (BSLAB) {
            b_voff[it] = u < B_UNITS ? (unsigned)(u * 16) : OOB;
        } else {
            const int row = stage_row(u >> 2, B_ROWS), part = u & 3;
            const int kx = row / BN, n = row - kx * BN;
            b_voff[it] = u < B_UNITS ? (unsigned)((kx * a.Nout + n) * Cin * 4 + part * 16) : OOB;
        }
    }

    auto decode = [&](int t, int& nb, int& img, int& y0, int& x0) {
        nb = t % NB;
        int r = t / NB;
        const int tx = r % a.tiles_x;
        r /= a.tiles_x;
        const int ty = r % a.tiles_y;
        img = r / a.tiles_y;
        y0 = ty * TH; x0 = tx * TW;
    };
    auto setup_load = [&](int t) {
        int nb, y0, x0;
        decode(t, nb, l_img, y0, x0);
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            const int u = tid + it * 256;
            const int hp = stage_row(u >> 2, A_PIX), part = u & 3;
            const int hy = hp / (TW + 2), hx = hp - hy * (TW + 2);
            const int gy = y0 + hy - 1, gx = x0 + hx - 1;
            const bool ok = u < A_UNITS && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
            a_voff[it] = ok ? (unsigned)(gy * a.W + gx) * (unsigned)((ELD_DBG(a) & 32) ? 64 : Cs0 * 4) + (unsigned)part * 16u : OOB;      // (dev probe 32: a chunk-planar tensor's addresses -- wrong data, full-line fetches)
        }
    };
    auto load_A = [&](int c0) {
        if (ELD_DBG(a) & 4) return;              // (dev ablation, compiled out of production builds: no staging loads)
        const char* src = static_cast<const char*>(c0 < a.C0 ? a.in0 : a.in1);
        const int cs = c0 < a.C0 ? c0 : c0 - a.C0;
        const size_t img_bytes = (size_t)a.H * a.W * Cs0 * 4;
        const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc((void*)(src + (size_t)l_img * img_bytes), 0, (int)img_bytes, 0x00020000);
        const int so = (ELD_DBG(a) & 32) ? (cs / 16) * a.H * a.W * 64 : cs * 4;
#pragma unroll
        for (int it = 0; it < A_IT; ++it)
            ra[it] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, (int)a_voff[it], so, 0));
    };
    auto load_B = [&](int nb, int c0, int ky) {
        if (ELD_DBG(a) & 4) return;
        const int wsoff = BSLAB ? ((ky * (Cin / CK) + c0 / CK) * NB + nb) * SLAB_BYTES : ((ky * 3 * a.Nout + nb * BN) * Cin + c0) * 4;
#pragma unroll
        for (int it = 0; it < B_IT; ++it)
            rb[it] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, (int)b_voff[it], wsoff, 0));
    };
    auto store_A = [&]() {
        if (ELD_DBG(a) & 16) return;             // (dev ablation: no halo cut / LDS stores)
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            const int u = tid + it * 256;
            if (u < A_UNITS) split_store(ldsA + stage_row(u >> 2, A_PIX) * PX + (u & 3) * 2, ra[it]);
        }
    };
    auto store_B = [&]() {
        if (ELD_DBG(a) & 8) return;              // (dev ablation: no slab stores)
#pragma unroll
        for (int it = 0; it < B_IT; ++it) {
            const int u = tid + it * 256;
            if constexpr (BSLAB) { if (u < B_UNITS) *reinterpret_cast<float4*>(ldsB + u * 4) = rb[it]; }      // the slab image as it is: consecutive lanes, consecutive 16 bytes
            else if (u < B_UNITS) split_store(ldsB + stage_row(u >> 2, B_ROWS) * PX + (u & 3) * 2, rb[it]);
        }
    };

    int t = xcd_block(a.xcd);
    if (t >= total_tiles) return;
    if ((ELD_DBG(a) & 256) && blockIdx.x >= (gridDim.x >> 1)) {      // (dev probe 256: the second workgroup of a CU starts half a tile period late)
        for (int i = 0; i < 4; ++i) __builtin_amdgcn_s_sleep(127);
    }
    setup_load(t);
    load_A(0);
    {
        int nb0, i0, y00, x00;
        decode(t, nb0, i0, y00, x00);
        load_B(nb0, 0, 0);
    }
    for (;;) {
        int nb, img, y0, x0;
        decode(t, nb, img, y0, x0);
        const int t_next = t + gridDim.x;
        f32x16 acc[RPW][NT];
#pragma unroll
        for (int r = 0; r < RPW; ++r)
#pragma unroll
            for (int tt = 0; tt < NT; ++tt)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[r][tt][i] = 0.f;

        for (int c0 = 0; c0 < Cin; c0 += CK) {
            const bool last_chunk = c0 + CK >= Cin;
#pragma unroll BFIRST ? 3 : 1            // (BFIRST: three explicit stages, so that every wait sees a static queue -- see the issue order below)
            for (int ky = 0; ky < 3; ++ky) {
                __builtin_amdgcn_s_setprio(3);   // the short staging section goes ahead of the co-resident workgroup's MFMA stream
                if (!(ELD_DBG(a) & 512)) __syncthreads();                 // every wave is done with the previous stage's operands  (dev probe 512: no barriers)
                if (ky == 0) store_A();
                store_B();
                if (!(ELD_DBG(a) & 512)) __syncthreads();
                {
                    // Order matters: vmcnt retires in order, so the loads the NEXT stage needs (its weight slab) go first and the halo tile -- needed
                    // three stages from now -- behind them: the compiler's wait for the slab registers then leaves the halo loads in flight (round 5;
                    // with the halo first, the next stage's wait for the slab drained the halo as well: one stage of latency instead of two)
                    auto issue_B = [&]() {
                        if (ky < 2) load_B(nb, c0, ky + 1);
                        else if (!last_chunk) load_B(nb, c0 + CK, 0);
                        else if (t_next < total_tiles) load_B(t_next % NB, 0, 0);
                    };
                    if constexpr (BFIRST) issue_B();          // (a compile-time choice: the compiler's wait counts follow the issue order)
                    if (ky == 0) {               // the halo tile of the next chunk / next tile has three stages to arrive
                        if (!last_chunk) load_A(c0 + CK);
                        else if (t_next < total_tiles) { setup_load(t_next); load_A(0); }
                    }
                    if constexpr (!BFIRST) issue_B();
                }
                __builtin_amdgcn_s_setprio(0);
                if (ELD_DBG(a) & 2) continue;    // (dev ablation: no fragment reads, no MFMAs)
                if constexpr (STREAM) {
                    x3_stage_blocks<RPW, NT>(acc,
                        [&](int kx, int r, uint4 (&X)[3]) {
                            const float* p = ldsA + ((wave * RPW + r + ky) * (TW + 2) + m + kx) * PX + hi * 4;
#pragma unroll
                            for (int pc = 0; pc < 3; ++pc) X[pc] = *reinterpret_cast<const uint4*>(p + pc * 8);
                        },
                        [&](int kx, int tt, uint4 (&Wt)[3]) {
                            const float* p = ldsB + (kx * BN + tt * 32 + m) * PX + hi * 4;
#pragma unroll
                            for (int pc = 0; pc < 3; ++pc) Wt[pc] = *reinterpret_cast<const uint4*>(p + pc * 8);
                        });
                    continue;
                }
                uint4 fx[DB ? 2 : 1][3][RPW], fw[DB ? 2 : 1][3][NT];
                auto read_tap = [&](int kx, uint4 (&X)[3][RPW], uint4 (&Wt)[3][NT]) {
#pragma unroll
                    for (int r = 0; r < RPW; ++r) {
                        const float* p = ldsA + ((wave * RPW + r + ky) * (TW + 2) + m + kx) * PX + hi * 4;
#pragma unroll
                        for (int pc = 0; pc < 3; ++pc) X[pc][r] = *reinterpret_cast<const uint4*>(p + pc * 8);
                    }
#pragma unroll
                    for (int tt = 0; tt < NT; ++tt) {
                        const float* p = ldsB + (kx * BN + tt * 32 + m) * PX + hi * 4;
#pragma unroll
                        for (int pc = 0; pc < 3; ++pc) Wt[pc][tt] = *reinterpret_cast<const uint4*>(p + pc * 8);
                    }
                };
                constexpr int WI[6] = {0, 1, 2, 0, 1, 0};
                constexpr int XI[6] = {2, 1, 0, 1, 0, 0};
                if constexpr (DB) {
                    read_tap(0, fx[0], fw[0]);
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const int cur = kx & 1;
                        if (kx + 1 < 3) read_tap(kx + 1, fx[cur ^ 1], fw[cur ^ 1]);
                        __builtin_amdgcn_sched_barrier(0);
                        // six piece products per (row, channel block), smallest first.  The pixel operand is held for up to six
                        // consecutive MFMAs (fewer operand-bus toggles: these kernels run at the power limit) while the two
                        // accumulators of the row alternate.
#pragma unroll
                        for (int r = 0; r < RPW; ++r)
#pragma unroll
                            for (int j = 2; j >= 0; --j)
#pragma unroll
                                for (int i = 0; i + j <= 2; ++i)
#pragma unroll
                                    for (int tt = 0; tt < NT; ++tt)
                                        acc[r][tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fw[cur][i][tt]),
                                                                                             __builtin_bit_cast(bf16x8, fx[cur][j][r]), acc[r][tt], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                } else {                         // register budget: one fragment set, the compiler interleaves reads and MFMAs
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        read_tap(kx, fx[0], fw[0]);
#pragma unroll
                        for (int q = 0; q < 6; ++q)
#pragma unroll
                            for (int r = 0; r < RPW; ++r)
#pragma unroll
                                for (int tt = 0; tt < NT; ++tt)
                                    acc[r][tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fw[0][WI[q]][tt]),
                                                                                         __builtin_bit_cast(bf16x8, fx[0][XI[q]][r]), acc[r][tt], 0, 0, 0);
                    }
                }
            }
        }

        // ---- epilogue: identical to conv_igemm.hip's (lane (m, hi) owns pixel x0+m and channels 8q+4hi..+3 of each 32-block)
        if (!(ELD_DBG(a) & 1)) {
            // stores in the full-line layout (conv.h f32_line_store: 16 pixels x 64 contiguous bytes per instruction); bias / saved activations are
            // read in the MFMA layout (own pixel m); every lane computes and takes part in the exchange, only loads and stores are predicated
            const int x = x0 + m;
            const bool xok = x < a.W;
            // forward: bias and max(0.2 v, v) once, in place (packed fp32 add / multiply, v_max without the canonicalising copy fmaxf() gets);
            // the pooled copy below reuses the finished values
            if (a.epi == EPI_FWD) {
                const float sl = a.lrelu ? 0.2f : 1.0f;                  // max(1 v, v) = v
#pragma unroll
                for (int tt = 0; tt < NT; ++tt)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 bq = *reinterpret_cast<const float4*>(a.bias + nb * BN + tt * 32 + 4 * hi + 8 * q);
#pragma unroll
                        for (int r = 0; r < RPW; ++r) bias_lrelu4(acc[r][tt], 4 * q, bq, sl);
                    }
                if (a.codes_out != nullptr && xok) {                     // slope codes of the finished values, for the backward-data epilogue that will want them
#pragma unroll
                    for (int r = 0; r < RPW; ++r) {
                        const int y = y0 + wave * RPW + r;
                        if (y >= a.H) continue;
                        const size_t pix = (size_t)(img * a.H + y) * a.W + x;
#pragma unroll
                        for (int tt = 0; tt < NT; ++tt)
                            a.codes_out[(pix * (size_t)(a.Nout >> 5) + (size_t)(nb * NT + tt)) * 2 + hi] = slope_codes16(acc[r][tt]);
                    }
                }
            }
            unsigned cw[RPW][NT];                                      // EPI_GRAD with slope codes: the whole tile's words, one load each, issued together
            if (a.epi != EPI_FWD && (a.codes0 != nullptr || a.codes1 != nullptr)) {
#pragma unroll
                for (int r = 0; r < RPW; ++r) {
                    const int y = y0 + wave * RPW + r;
                    const size_t pix = (size_t)(img * a.H + (y < a.H ? y : 0)) * a.W + (xok ? x : 0);
#pragma unroll
                    for (int tt = 0; tt < NT; ++tt) {
                        const int nb32 = nb * BN + tt * 32;
                        const bool lo = nb32 < a.split;
                        const unsigned* cd = lo ? a.codes0 : a.codes1;
                        const int C = lo ? a.split : a.Nout - a.split, cb = lo ? nb32 : nb32 - a.split;
                        cw[r][tt] = cd != nullptr ? cd[(pix * (size_t)(C >> 5) + (size_t)(cb >> 5)) * 2 + hi] : 0u;
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                const int y = y0 + wave * RPW + r;                      // wave-uniform
                const bool yok = y < a.H;
                const size_t rowpix = (size_t)(img * a.H + (yok ? y : 0)) * a.W;
                const size_t pix = rowpix + (xok ? x : 0);
#pragma unroll
                for (int tt = 0; tt < NT; ++tt) {
                    const int nb32 = nb * BN + tt * 32;
                    float4 v[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = make_float4(acc[r][tt][4 * q], acc[r][tt][4 * q + 1], acc[r][tt][4 * q + 2], acc[r][tt][4 * q + 3]);
                    float* blk;
                    int C;
                    if (a.epi == EPI_FWD) {
                        C = a.Nout;
                        blk = static_cast<float*>(a.out0) + (rowpix + x0) * C + nb32;
                    } else {                                            // a 32-channel block never straddles the concat split
                        const bool lo = nb32 < a.split;
                        C = lo ? a.split : a.Nout - a.split;
                        const int cb = lo ? nb32 : nb32 - a.split;
                        blk = static_cast<float*>(lo ? a.out0 : a.out1) + (rowpix + x0) * C + cb;
                        const float* act = static_cast<const float*>(lo ? a.act0 : a.act1);
                        if ((lo ? a.codes0 : a.codes1) != nullptr) {
                            const unsigned w = cw[r][tt];
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                v[q].x *= slope_of_code(w, 4 * q); v[q].y *= slope_of_code(w, 4 * q + 1);
                                v[q].z *= slope_of_code(w, 4 * q + 2); v[q].w *= slope_of_code(w, 4 * q + 3);
                            }
                        } else if (act != nullptr) {
                            float4 s[4];
#pragma unroll
                            for (int q = 0; q < 4; ++q) s[q] = *reinterpret_cast<const float4*>(act + pix * C + cb + 4 * hi + 8 * q);
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                v[q].x *= lrelu_slope(s[q].x); v[q].y *= lrelu_slope(s[q].y);
                                v[q].z *= lrelu_slope(s[q].z); v[q].w *= lrelu_slope(s[q].w);
                            }
                        }
                    }
                    f32_line_store(v, blk, (size_t)C, lane, yok, a.W - x0, (ELD_DBG(a) & 128) != 0);      // (dev probe 128: write-through stores)
                }
            }
        }
        if (a.epi == EPI_FWD && a.pool_out != nullptr && !(ELD_DBG(a) & 1)) {
            int img_p[RPW / 2], y_p[RPW / 2];
#pragma unroll
            for (int rp = 0; rp < RPW / 2; ++rp) { img_p[rp] = img; y_p[rp] = y0 + wave * RPW + 2 * rp; }
            pool_epilogue<RPW, NT, BN, false>(a, acc, img_p, nb, y_p, x0 + m, hi);      // (no pool codes from this kernel: unet.hip asks for them only where conv_x3w_kernel runs)
        }
        if (t_next >= total_tiles) break;
        t = t_next;
    }
}

// =============================================================================================================================
// conv_x3d_kernel: the same convolution with the WEIGHT operand pre-split once per step (pack kernels, unet_misc.hip) and laid out
// in global memory as the exact LDS image of a stage's slab, so that a stage's weights travel L2 -> LDS by LDS-DMA
// (buffer_load_dwordx4 ... lds: no VGPRs, no split VALU, no ds_write) into a double buffer, one stage ahead of their use.
// Only the activation halo tile still goes through registers (it is fp32 in HBM and must be cut into pieces on the way in),
// once per 16-channel chunk = once per three stages.
//
//   packed weights:  slab(ky, chunk, nb) = [kx 0..2][n 0..BN)[3 pieces][16 bf16] + 16 B pad per row  (3*BN rows x 112 B, 1 KiB-multiple)
//                    at byte offset ((ky * (K/16) + chunk) * (Nout/BN) + nb) * 3*BN*112          (x3_store in unet_misc.hip)
//   stage s:         barrier [slab s landed (each wave drained its own DMA pieces before arriving) / everybody done with stage s-1]
//                    ky == 0:  cut the halo registers into LDS, barrier
//                    issue DMA of slab s+1 into the other buffer; ky == 0: issue the next chunk's halo loads into registers
//                    3 taps x 6 piece products x RPW x NT MFMAs out of LDS
// Barriers per chunk: 4 (was 6); VALU per MFMA: the activation cut only.  WAVES = 4 (two workgroups per CU) or 8 (one).
// One 1 KiB LDS-DMA piece: lane l copies the 16 bytes at buffer offset voff (per lane) + soff (wave-uniform) to LDS byte address
// lds_dst + 16 l (lds_dst wave-uniform).  Issued through inline asm ON PURPOSE: hipcc orders every later ds_read behind an LDS-DMA it
// knows about (s_waitcnt vmcnt(0) in front of the stage's first fragment read, which would serialise the prefetch and drain the halo
// loads with it).  The kernel waits for its own pieces explicitly (dma_wait) before the barrier that publishes the slab.
// s_nop 4: SALU-written soffset / descriptor -> VMEM read wait states; s_nop 0: M0 write -> LDS-DMA; M0 is saved and restored.
template <int BN, int RPW, int WAVES, bool DB, int STREAM = 0>       // STREAM: 0 = the round-5 loops, 1 = streamed blocks, 2 = streamed blocks in the weight-reuse order (conv_x3_dev.h)
__global__ __launch_bounds__(64 * WAVES, (WAVES == 4 && BN <= 64) ? 2 : (WAVES == 8 ? 2 : 1)) void conv_x3d_kernel(const ConvArgs a) {
    constexpr int CK = 16, THREADS = 64 * WAVES;
    constexpr int TH = WAVES * RPW, NT = BN / 32, A_PIX = (TH + 2) * (TW + 2);
    constexpr int B_ROWS = 3 * BN;
    constexpr bool CODES = NT <= 2;                                     // slope codes (ConvArgs::codes_out / codes0 / codes1) are honoured by the 32- / 64-channel tiles
    constexpr int B_WORDS = (B_ROWS * PX * 4 + 1023) / 1024 * 256;     // slab stride (global and LDS): rows padded to whole 1 KiB DMA pieces
    constexpr int B_PIECES = B_WORDS * 4 / 1024;                      // 1 KiB per wave-instruction
    constexpr int DMA_IT = (B_PIECES + WAVES - 1) / WAVES;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* ldsB = lds;                                                // two slabs first: the DMA destinations stay at low LDS addresses
    float* ldsA = lds + 2 * B_WORDS;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 31, hi = lane >> 5;
    const int NB = a.Nout / BN;
    const int Cin = a.C0 + a.C1, NCH = Cin / CK;
    // work item = (tile, K part): with a.ksplit > 1 (small problems) the chunk range of a tile is shared out over ksplit workgroups that write raw
    // partial sums; x3_splitk_finish_kernel adds them in a fixed order and runs the epilogue
    constexpr bool SPLITK = WAVES == 4;                                     // only the small-tile variant carries the split (the 8-wave kernels have no register to spare)
    const int KS = SPLITK && a.ksplit > 1 ? a.ksplit : 1;
    const int total_tiles = a.tiles_x * a.tiles_y * NB * KS;                // work items; tiles_y: tile rows of the virtual strip (all images)
    const int Cs0 = a.C0;
    // Round 4 (see conv_bfd.hip): the workgroup's TH * 32 pixel slots are the pixels of a tile_h x tile_w tile of the launch's choosing
    // (slot p = tile pixel (p / tile_w, p % tile_w)), rows are rows of the virtual strip of the batch (conv.h vrow_*).
    const int THL = a.tile_h, TWL = a.tile_w, HWL = TWL + 2;
    const int APX = (THL + 2) * HWL, NPX = THL * TWL, VP = a.vp;
    const bool seam = VP % THL != 0;
    const bool grouped = conv_slots_grouped(THL, TWL, TH);                  // slots numbered lane group by lane group (conv.h conv_tile_shape)
    const unsigned mag_hw = (unsigned)((0x100000000ull + (unsigned)HWL - 1) / (unsigned)HWL);      // x / HWL == umulhi(x, mag_hw) for x < 2^16
    const unsigned mag_tw = (unsigned)((0x100000000ull + (unsigned)TWL - 1) / (unsigned)TWL);

    constexpr int A_UNITS = A_PIX * 4;
    constexpr int A_IT = (A_UNITS + THREADS - 1) / THREADS;
    float4 ra[A_IT];
    constexpr unsigned OOB = 0xFFFFFFF0u;
    unsigned a_voff[A_IT];
    int l_img = 0;
    const unsigned ldsB_addr = (unsigned)__builtin_amdgcn_readfirstlane((int)(size_t)(__attribute__((address_space(3))) float*)ldsB);
    const unsigned long long wbase = (unsigned long long)a.wp;
    const i32x4 rsrc_w = {(int)(unsigned)wbase, (int)((unsigned)(wbase >> 32) & 0xFFFFu), (int)((size_t)3 * NCH * NB * B_WORDS * 4), 0x00020000};
    const unsigned dma_voff = (unsigned)lane * 16u;

    // tile -> channel block, image of the tile's first strip row (img0), that row's offset in the image's pitch (vrel), first column
    auto decode = [&](int t, int& nb, int& img0, int& vrel, int& x0) {
        nb = t % NB;
        const int r = t / NB;
        const int ty = r / a.tiles_x, tx = r - ty * a.tiles_x;
        const int v0 = ty * THL;
        img0 = v0 / VP; vrel = v0 - img0 * VP;
        x0 = tx * TWL;
    };
    // strip row relative to image img0's pitch -> (image offset 0 / 1, row); false: a separator row, or beyond the strip
    auto strip_row = [&](int row, int& dimg, int& y) -> bool {
        const bool over = row >= VP;
        dimg = over ? 1 : 0;
        y = over ? row - VP : row;
        return over ? seam && y < a.H : (unsigned)row < (unsigned)a.H;
    };
    auto setup_load = [&](int t) {
        int nb, vrel, x0;
        decode(t, nb, l_img, vrel, x0);
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            const int u = tid + it * THREADS;
            const int hp = stage_row(u >> 2, A_PIX), part = u & 3;
            const int hy = (int)__umulhi((unsigned)hp, mag_hw), hx = hp - hy * HWL;
            int dimg, y;
            const bool rok = strip_row(vrel + hy - 1, dimg, y);
            const int gx = x0 + hx - 1;
            const bool ok = u < A_UNITS && hp < APX && rok && gx >= 0 && gx < a.W;      // rows of a second image that does not exist: outside the descriptor
            a_voff[it] = ok ? (unsigned)((dimg * a.H + y) * a.W + gx) * (unsigned)((ELD_DBG(a) & 32) ? 64 : Cs0 * 4) + (unsigned)part * 16u : OOB;      // (dev probe 32, as in conv_x3_kernel)
        }
    };
    auto load_A = [&](int c0) {
        if (ELD_DBG(a) & 4) return;              // (dev ablation, compiled out of production builds: no halo loads)
        const char* src = static_cast<const char*>(c0 < a.C0 ? a.in0 : a.in1);
        const int cs = c0 < a.C0 ? c0 : c0 - a.C0;
        const size_t img_bytes = (size_t)a.H * a.W * Cs0 * 4;
        const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc((void*)(src + (size_t)l_img * img_bytes), 0,
                                                                                (int)(unsigned)(img_bytes * (size_t)(a.N - l_img < 2 ? a.N - l_img : 2)), 0x00020000);
        const int so = (ELD_DBG(a) & 32) ? (cs / 16) * a.H * a.W * 64 : cs * 4;
#pragma unroll
        for (int it = 0; it < A_IT; ++it)
            ra[it] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, (int)a_voff[it], so, 0));
    };
    auto store_A = [&]() {
        if (ELD_DBG(a) & 16) return;             // (dev ablation: no halo cut / LDS stores)
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            const int u = tid + it * THREADS;
            if (u < A_UNITS) split_store(ldsA + stage_row(u >> 2, A_PIX) * PX + (u & 3) * 2, ra[it]);
        }
    };
    // slab (nb, chunk, ky) -> LDS buffer `buf`: wave w moves pieces w, w + WAVES, ...
    auto dma_B = [&](int buf, int nb, int chunk, int ky) {
        if (ELD_DBG(a) & 8) return;              // (dev ablation: no slab DMA)
        const unsigned soff = (unsigned)(((ky * NCH + chunk) * NB + nb) * (B_WORDS * 4));
#pragma unroll
        for (int it = 0; it < DMA_IT; ++it) {
            const int piece = wave + it * WAVES;             // wave-uniform
            if (piece < B_PIECES) bdma16(rsrc_w, dma_voff, soff + (unsigned)(piece * 1024), ldsB_addr + (unsigned)(buf * B_WORDS * 4 + piece * 1024));
        }
    };

    // LDS word offset of this lane's pixel slot in MFMA column r (+ its k half): slot p = (wave * RPW + r) * 32 + m -> halo pixel (p / TWL, p % TWL)
    int pb[RPW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        int p = (wave * RPW + r) * 32 + conv_slot_of_lane(m, grouped);
        p = p < NPX ? p : 0;
        const int tr = (int)__umulhi((unsigned)p, mag_tw);
        pb[r] = (tr * HWL + (p - tr * TWL)) * PX + hi * 4;
    }
    auto chunk_begin = [&](int t) { return ((t % KS) * NCH) / KS; };
    auto chunk_end = [&](int t) { return ((t % KS + 1) * NCH) / KS; };
    int t = xcd_block(a.xcd);
    if (t >= total_tiles) return;
    setup_load(t / KS);
    load_A(chunk_begin(t) * CK);
    int buf = 0;
    bool after_epi = false;
    {
        int nb0, i0, v00, x00;
        decode(t / KS, nb0, i0, v00, x00);
        dma_B(0, nb0, chunk_begin(t), 0);
    }
    for (;;) {
        int nb, img0, vrel, x0;
        decode(t / KS, nb, img0, vrel, x0);
        const int t_next = t + gridDim.x;
        const int c_begin = chunk_begin(t), c_end = chunk_end(t);
        f32x16 acc[RPW][NT];
#pragma unroll
        for (int r = 0; r < RPW; ++r)
#pragma unroll
            for (int tt = 0; tt < NT; ++tt)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[r][tt][i] = 0.f;

        for (int chunk = c_begin; chunk < c_end; ++chunk) {
            const bool last_chunk = chunk + 1 >= c_end;
#pragma unroll 1
            for (int ky = 0; ky < 3; ++ky) {
                __builtin_amdgcn_s_setprio(3);   // the short staging section goes ahead of co-resident waves' MFMA streams
                if ((ELD_DBG(a) & 2048) && after_epi) after_epi = false;      // (dev probe 2048: no drain of the epilogue's stores -- timing only, the slab is not waited for)
                else
                dma_wait();                      // this wave's pieces of the stage's slab (issued one stage ago) have landed (round 5: a counted wait that
                                                 // leaves the halo loads of the chunk's first stage in flight here measured 0.0 %: profiles/r05_ab_notes.md)
                __syncthreads();                 // ... and so have everybody else's; the previous stage's fragment reads are done
                if (ky == 0) {
                    store_A();
                    __syncthreads();
                }
                if (ky < 2) dma_B(buf ^ 1, nb, chunk, ky + 1);
                else if (!last_chunk) dma_B(buf ^ 1, nb, chunk + 1, 0);
                else if (t_next < total_tiles) dma_B(buf ^ 1, (t_next / KS) % NB, chunk_begin(t_next), 0);
                if (ky == 0) {                   // the halo tile of the next chunk / next tile has three stages to arrive
                    if (!last_chunk) load_A((chunk + 1) * CK);
                    else if (t_next < total_tiles) { setup_load(t_next / KS); load_A(chunk_begin(t_next) * CK); }
                }
                __builtin_amdgcn_s_setprio(0);
                if (ELD_DBG(a) & 2) { buf ^= 1; continue; }      // (dev ablation: no fragment reads, no MFMAs)
                const float* lb = ldsB + buf * B_WORDS;
                if constexpr (STREAM) {
                    x3_stage_blocks<RPW, NT, STREAM == 2>(acc,
                        [&](int kx, int r, uint4 (&X)[3]) {
                            const float* p = ldsA + pb[r] + (ky * HWL + kx) * PX;
#pragma unroll
                            for (int pc = 0; pc < 3; ++pc) X[pc] = *reinterpret_cast<const uint4*>(p + pc * 8);
                        },
                        [&](int kx, int tt, uint4 (&Wt)[3]) {
                            const float* p = lb + (kx * BN + tt * 32 + m) * PX + hi * 4;
#pragma unroll
                            for (int pc = 0; pc < 3; ++pc) Wt[pc] = *reinterpret_cast<const uint4*>(p + pc * 8);
                        });
                    buf ^= 1;
                    continue;
                }
                uint4 fx[DB ? 2 : 1][3][RPW], fw[DB ? 2 : 1][3][NT];
                auto read_tap = [&](int kx, uint4 (&X)[3][RPW], uint4 (&Wt)[3][NT]) {
#pragma unroll
                    for (int r = 0; r < RPW; ++r) {
                        const float* p = ldsA + pb[r] + (ky * HWL + kx) * PX;
#pragma unroll
                        for (int pc = 0; pc < 3; ++pc) X[pc][r] = *reinterpret_cast<const uint4*>(p + pc * 8);
                    }
#pragma unroll
                    for (int tt = 0; tt < NT; ++tt) {
                        const float* p = lb + (kx * BN + tt * 32 + m) * PX + hi * 4;
#pragma unroll
                        for (int pc = 0; pc < 3; ++pc) Wt[pc][tt] = *reinterpret_cast<const uint4*>(p + pc * 8);
                    }
                };
                constexpr int WI[6] = {0, 1, 2, 0, 1, 0};
                constexpr int XI[6] = {2, 1, 0, 1, 0, 0};
                if constexpr (DB) {
                    read_tap(0, fx[0], fw[0]);
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const int cur = kx & 1;
                        if (kx + 1 < 3) read_tap(kx + 1, fx[cur ^ 1], fw[cur ^ 1]);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int r = 0; r < RPW; ++r)
#pragma unroll
                            for (int j = 2; j >= 0; --j)
#pragma unroll
                                for (int i = 0; i + j <= 2; ++i)
#pragma unroll
                                    for (int tt = 0; tt < NT; ++tt)
                                        acc[r][tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fw[cur][i][tt]),
                                                                                             __builtin_bit_cast(bf16x8, fx[cur][j][r]), acc[r][tt], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                } else {
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        read_tap(kx, fx[0], fw[0]);
#pragma unroll
                        for (int q = 0; q < 6; ++q)
#pragma unroll
                            for (int r = 0; r < RPW; ++r)
#pragma unroll
                                for (int tt = 0; tt < NT; ++tt)
                                    acc[r][tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fw[0][WI[q]][tt]),
                                                                                         __builtin_bit_cast(bf16x8, fx[0][XI[q]][r]), acc[r][tt], 0, 0, 0);
                    }
                }
                buf ^= 1;
            }
        }

        // ---- epilogue: lane (m, hi) owns pixel slot m of MFMA column r and channels 8q+4hi..+3 of each 32-block (as conv_x3_kernel) ----
        // slot of column r -> NHW pixel index; false: no such pixel (slot beyond the tile, separator row, outside the image)
        auto slot_pixel = [&](int r, int col, size_t& pix) -> bool {       // col: MFMA column (lane & 31 of the lane that computed it)
            const int p = (wave * RPW + r) * 32 + conv_slot_of_lane(col, grouped);
            const int tr = (int)__umulhi((unsigned)p, mag_tw), x = x0 + p - tr * TWL;
            int dimg, y;
            const bool ok = strip_row(vrel + tr, dimg, y) && p < NPX && img0 + dimg < a.N && x < a.W;
            pix = ok ? (size_t)((img0 + dimg) * a.H + y) * a.W + x : 0;
            return ok;
        };
        if (ELD_DBG(a) & 1) {                    // (dev ablation: no epilogue)
        } else if (KS > 1) {                     // split K: raw partial sums of this K part, [part][image][y][x][Nout]
            float* pbase = a.kpart + (size_t)(t % KS) * ((size_t)a.N * a.H * a.W * a.Nout);
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                size_t pix;
                if (!slot_pixel(r, m, pix)) continue;
                float* prow = pbase + pix * a.Nout + nb * BN + 4 * hi;
#pragma unroll
                for (int tt = 0; tt < NT; ++tt)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        *reinterpret_cast<float4*>(prow + tt * 32 + 8 * q) = make_float4(acc[r][tt][4 * q], acc[r][tt][4 * q + 1], acc[r][tt][4 * q + 2], acc[r][tt][4 * q + 3]);
            }
        } else {
            // stores in the full-line layout (conv.h f32_line_store2: 16 slots x 64 contiguous bytes per instruction); bias / saved activations are
            // read in the MFMA layout (own slot m); every lane computes and takes part in the exchange, only loads and stores are predicated
            // forward: bias and max(0.2 v, v) once, in place (packed fp32 add / multiply, v_max without the canonicalising copy fmaxf() gets);
            // the pooled copy below reuses the finished values
            if (a.epi == EPI_FWD) {
                const float sl = a.lrelu ? 0.2f : 1.0f;                  // max(1 v, v) = v
#pragma unroll
                for (int tt = 0; tt < NT; ++tt)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 bq = *reinterpret_cast<const float4*>(a.bias + nb * BN + tt * 32 + 4 * hi + 8 * q);
#pragma unroll
                        for (int r = 0; r < RPW; ++r) bias_lrelu4(acc[r][tt], 4 * q, bq, sl);
                    }
                // slope codes of the finished values (conv.h ConvArgs::codes_out); the 32- / 64-channel tiles only: the 128-channel tile has no register to spare
                // (its instantiation spills 40 bytes more with this code, checked offline) and its launches gain least (34 ... 173 us each)
                if (CODES && a.codes_out != nullptr) {
#pragma unroll
                    for (int r = 0; r < RPW; ++r) {
                        size_t pix;
                        if (!slot_pixel(r, m, pix)) continue;
#pragma unroll
                        for (int tt = 0; tt < NT; ++tt)
                            a.codes_out[(pix * (size_t)(a.Nout >> 5) + (size_t)(nb * NT + tt)) * 2 + hi] = slope_codes16(acc[r][tt]);
                    }
                }
            }
            const bool coded = CODES && a.epi != EPI_FWD && (a.codes0 != nullptr || a.codes1 != nullptr);
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                size_t pix, pix0, pix1;
                slot_pixel(r, m, pix);                                  // own slot (0 where there is none: the load re-reads a valid pixel)
                const bool ok0 = slot_pixel(r, lane & 15, pix0), ok1 = slot_pixel(r, (lane & 15) + 16, pix1);
                unsigned cw[NT];                                        // slope-code words of this row's channel blocks: one load each, issued together
                if (coded) {
#pragma unroll
                    for (int tt = 0; tt < NT; ++tt) {
                        const int nb32 = nb * BN + tt * 32;
                        const bool lo = nb32 < a.split;
                        const unsigned* cd = lo ? a.codes0 : a.codes1;
                        const int C = lo ? a.split : a.Nout - a.split, cb = lo ? nb32 : nb32 - a.split;
                        cw[tt] = cd != nullptr ? cd[(pix * (size_t)(C >> 5) + (size_t)(cb >> 5)) * 2 + hi] : 0u;
                    }
                }
#pragma unroll
                for (int tt = 0; tt < NT; ++tt) {
                    const int nb32 = nb * BN + tt * 32;
                    float4 v[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = make_float4(acc[r][tt][4 * q], acc[r][tt][4 * q + 1], acc[r][tt][4 * q + 2], acc[r][tt][4 * q + 3]);
                    float* blk;                                         // channel 0 of the block in pixel 0 of the destination tensor
                    int C;
                    if (a.epi == EPI_FWD) {
                        C = a.Nout;
                        blk = static_cast<float*>(a.out0) + nb32;
                    } else {                                            // a 32-channel block never straddles the concat split
                        const bool lo = nb32 < a.split;
                        C = lo ? a.split : a.Nout - a.split;
                        const int cb = lo ? nb32 : nb32 - a.split;
                        blk = static_cast<float*>(lo ? a.out0 : a.out1) + cb;
                        const float* act = static_cast<const float*>(lo ? a.act0 : a.act1);
                        if (CODES && (lo ? a.codes0 : a.codes1) != nullptr) {
                            const unsigned w = cw[tt];
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                v[q].x *= slope_of_code(w, 4 * q); v[q].y *= slope_of_code(w, 4 * q + 1);
                                v[q].z *= slope_of_code(w, 4 * q + 2); v[q].w *= slope_of_code(w, 4 * q + 3);
                            }
                        } else if (act != nullptr) {
                            float4 s[4];
#pragma unroll
                            for (int q = 0; q < 4; ++q) s[q] = *reinterpret_cast<const float4*>(act + pix * C + cb + 4 * hi + 8 * q);
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                v[q].x *= lrelu_slope(s[q].x); v[q].y *= lrelu_slope(s[q].y);
                                v[q].z *= lrelu_slope(s[q].z); v[q].w *= lrelu_slope(s[q].w);
                            }
                        }
                    }
                    f32_line_store2(v, ok0 ? blk + pix0 * C : nullptr, ok1 ? blk + pix1 * C : nullptr, lane, (ELD_DBG(a) & 128) != 0);
                }
            }
        }
        if (KS == 1 && a.epi == EPI_FWD && a.pool_out != nullptr && !(ELD_DBG(a) & 1)) {     // pooled launches run TH x 32 tiles (launcher): slot m of column r = tile pixel (wave * RPW + r, m)
            int img_p[RPW / 2], y_p[RPW / 2];                           // (a lane's row pairs may lie on both sides of a seam of the strip)
#pragma unroll
            for (int rp = 0; rp < RPW / 2; ++rp) {
                int dimg, y;
                const bool rok = strip_row(vrel + wave * RPW + 2 * rp, dimg, y) && img0 + dimg < a.N;
                img_p[rp] = img0 + dimg; y_p[rp] = rok ? y : a.H;
            }
            pool_epilogue<RPW, NT, BN, (WAVES == 8 && NT <= 2)>(a, acc, img_p, nb, y_p, x0 + m, hi);
        }
        if (t_next >= total_tiles) break;
        t = t_next;
        after_epi = true;
    }
}

// ---- transposed conv 2x2/s2 as a pixel GEMM on the same split-operand scheme -----------------------------------------------
// MODE CONV_1X1 (forward): out[(2y+dy, 2x+dx)][co] = sum_c in[(y,x)][c] * Wp[(dy,dx,co)][c]     (N = 4*Cout, K = Cin)
// MODE CONV_GATHER2X2 (backward-data): din[(y,x)][ci] = sum_{tap,c} dout[(2y+dy, 2x+dx)][c] * Wp[tap][ci][c]   (K = 4*Cout)
// Tile = 2*WAVES rows x 32 pixels x BN channels, K stage = 32 k-values (two 16-k sub-chunks; a stage never straddles a tap):
//   <64, 4>:  8 x 32 x 64,  LDS = [sub][256 pixels][28 words] + [sub][64][28 words] = 71.7 KB, two workgroups per CU;
//   <128, 8>: 16 x 32 x 128, 143 KB, one 8-wave workgroup per CU -- each cut operand feeds twice the MFMAs.
template <int MODE, int BN, int WAVES>
__global__ __launch_bounds__(64 * WAVES, WAVES == 4 ? 2 : 1) void conv_x3_gemm_kernel(const ConvArgs a) {
    constexpr int THREADS = 64 * WAVES, RPW = 2, TH = WAVES * RPW, NT = BN / 32, NS = 2;
    constexpr int TPIX = TH * TW;
    constexpr int A_WORDS = NS * TPIX * PX;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* ldsA = lds;
    float* ldsB = lds + A_WORDS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = lane & 31, hi = lane >> 5;
    const int NB = a.Nout / BN;
    const int C0 = a.C0;                                           // channels of the source tensor
    const int Ktot = MODE == CONV_1X1 ? C0 : 4 * C0;
    const int Ws = MODE == CONV_GATHER2X2 ? 2 * a.W : a.W, Hs = MODE == CONV_GATHER2X2 ? 2 * a.H : a.H;
    const int total_tiles = a.tiles_x * a.tiles_y * a.N * NB;

    constexpr int A_UNITS = NS * TPIX * 4, B_UNITS = NS * BN * 4;
    constexpr int A_IT = A_UNITS / THREADS, B_IT = B_UNITS / THREADS;
    static_assert(A_UNITS % THREADS == 0 && B_UNITS % THREADS == 0, "whole staging passes");
    float4 ra[A_IT], rb[B_IT];
    constexpr unsigned OOB = 0xFFFFFFF0u;
    unsigned a_voff[A_IT], b_voff[B_IT];
    int l_img = 0;
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc((void*)a.wp, 0, (int)((size_t)(MODE == CONV_1X1 ? 1 : 4) * a.Nout * C0 * 4), 0x00020000);
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
        const int u = tid + it * THREADS;                              // (sub, n, part)
        const int part = u & 3, rr = stage_row(u >> 2, NS * BN), n = rr % BN, sub = rr / BN;
        b_voff[it] = (unsigned)(n * C0 * 4 + sub * 64 + part * 16);
    }
    auto decode = [&](int t, int& nb, int& img, int& y0, int& x0) {
        nb = t % NB;
        int r = t / NB;
        const int tx = r % a.tiles_x;
        r /= a.tiles_x;
        const int ty = r % a.tiles_y;
        img = r / a.tiles_y;
        y0 = ty * TH; x0 = tx * TW;
    };
    auto setup_load = [&](int t) {
        int nb, y0, x0;
        decode(t, nb, l_img, y0, x0);
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            const int u = tid + it * THREADS;                          // (sub, pixel, part)
            const int part = u & 3, rr = stage_row(u >> 2, NS * TPIX), lp = rr % TPIX, sub = rr / TPIX;
            const int gy = y0 + lp / TW, gx = x0 + lp % TW;
            const bool ok = gy < a.H && gx < a.W;
            const unsigned pix = MODE == CONV_GATHER2X2 ? (unsigned)(2 * gy * Ws + 2 * gx) : (unsigned)(gy * Ws + gx);
            a_voff[it] = ok ? pix * (unsigned)(C0 * 4) + (unsigned)(sub * 64 + part * 16) : OOB;
        }
    };
    auto load_stage = [&](int nb, int k0) {
        const int tap = MODE == CONV_GATHER2X2 ? k0 / C0 : 0;
        const int cs = k0 - tap * C0;
        const size_t img_bytes = (size_t)Hs * Ws * C0 * 4;
        const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc((void*)(static_cast<const char*>(a.in0) + (size_t)l_img * img_bytes), 0, (int)img_bytes, 0x00020000);
        const int asoff = (((tap >> 1) * Ws + (tap & 1)) * C0 + cs) * 4;
#pragma unroll
        for (int it = 0; it < A_IT; ++it)
            ra[it] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, (int)a_voff[it], asoff, 0));
        const int wsoff = ((tap * a.Nout + nb * BN) * C0 + cs) * 4;
#pragma unroll
        for (int it = 0; it < B_IT; ++it)
            rb[it] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, (int)b_voff[it], wsoff, 0));
    };
    auto store_stage = [&]() {
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            const int u = tid + it * THREADS;
            split_store(ldsA + stage_row(u >> 2, NS * TPIX) * PX + (u & 3) * 2, ra[it]);          // row = sub * TPIX + pixel
        }
#pragma unroll
        for (int it = 0; it < B_IT; ++it) {
            const int u = tid + it * THREADS;
            split_store(ldsB + stage_row(u >> 2, NS * BN) * PX + (u & 3) * 2, rb[it]);
        }
    };

    int t = xcd_block(a.xcd);
    if (t >= total_tiles) return;
    setup_load(t);
    load_stage(t % NB, 0);
    for (;;) {
        int nb, img, y0, x0;
        decode(t, nb, img, y0, x0);
        const int t_next = t + gridDim.x;
        f32x16 acc[RPW][NT];
#pragma unroll
        for (int r = 0; r < RPW; ++r)
#pragma unroll
            for (int tt = 0; tt < NT; ++tt)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[r][tt][i] = 0.f;
        for (int k0 = 0; k0 < Ktot; k0 += 16 * NS) {
            __syncthreads();
            if (!(ELD_DBG(a) & 16)) store_stage();                    // (dev ablations, ELD_DEV_TOOLS builds: 16 no cut / LDS stores, 4 no staging loads, 2 no fragment reads / MFMAs, 1 no epilogue)
            __syncthreads();
            if (!(ELD_DBG(a) & 4)) {
            if (k0 + 16 * NS < Ktot) load_stage(nb, k0 + 16 * NS);
            else if (t_next < total_tiles) { setup_load(t_next); load_stage(t_next % NB, 0); }
            }
            if (ELD_DBG(a) & 2) continue;
            constexpr int WI[6] = {0, 1, 2, 0, 1, 0};
            constexpr int XI[6] = {2, 1, 0, 1, 0, 0};
#pragma unroll
            for (int sub = 0; sub < NS; ++sub) {
                constexpr int NTH = NT > 2 ? 1 : NT;                  // channel blocks whose fragments are live together (register budget at NT = 4)
                uint4 fx[3][RPW];
#pragma unroll
                for (int r = 0; r < RPW; ++r) {
                    const float* p = ldsA + (sub * TPIX + (wave * RPW + r) * TW + m) * PX + hi * 4;
#pragma unroll
                    for (int pc = 0; pc < 3; ++pc) fx[pc][r] = *reinterpret_cast<const uint4*>(p + pc * 8);
                }
#pragma unroll
                for (int t0 = 0; t0 < NT; t0 += NTH) {
                    uint4 fw[3][NTH];
#pragma unroll
                    for (int tt = 0; tt < NTH; ++tt) {
                        const float* p = ldsB + (sub * BN + (t0 + tt) * 32 + m) * PX + hi * 4;
#pragma unroll
                        for (int pc = 0; pc < 3; ++pc) fw[pc][tt] = *reinterpret_cast<const uint4*>(p + pc * 8);
                    }
#pragma unroll
                    for (int q = 0; q < 6; ++q)
#pragma unroll
                        for (int r = 0; r < RPW; ++r)
#pragma unroll
                            for (int tt = 0; tt < NTH; ++tt)
                                acc[r][t0 + tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fw[WI[q]][tt]), __builtin_bit_cast(bf16x8, fx[XI[q]][r]),
                                                                                          acc[r][t0 + tt], 0, 0, 0);
                    if constexpr (NT > NTH) __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        if (!(ELD_DBG(a) & 1))
        {   // epilogue: the MFMA leaves lane (m, hi) with channels 8q+4hi..+3 of pixel x0+m in each 32-block; stores in the full-line layout (conv.h f32_line_store)
            const int x = x0 + m;
            const bool xok = x < a.W;
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                const int y = y0 + wave * RPW + r;
                const bool yok = y < a.H;
                const int yc = yok ? y : 0;
                const size_t pix = (size_t)(img * a.H + yc) * a.W + (xok ? x : 0);
#pragma unroll
                for (int tt = 0; tt < NT; ++tt) {
                    const int nb32 = nb * BN + tt * 32, nbase = nb32 + 4 * hi;
                    float4 v[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = make_float4(acc[r][tt][4 * q], acc[r][tt][4 * q + 1], acc[r][tt][4 * q + 2], acc[r][tt][4 * q + 3]);
                    float* blk;
                    size_t pstride;
                    if (MODE == CONV_1X1) {                              // n = tap * Cout + co: a 32-block never straddles a tap (Cout % 32 == 0)
                        const int tap = nb32 / a.Cout_t, co0 = nb32 - tap * a.Cout_t;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float4 bq = *reinterpret_cast<const float4*>(a.bias + co0 + 4 * hi + 8 * q);
                            v[q].x += bq.x; v[q].y += bq.y; v[q].z += bq.z; v[q].w += bq.w;
                        }
                        blk = static_cast<float*>(a.out0) + ((size_t)(img * 2 * a.H + 2 * yc + (tap >> 1)) * (2 * a.W) + 2 * x0 + (tap & 1)) * a.Cout_t + co0;
                        pstride = (size_t)2 * a.Cout_t;
                    } else {
                        blk = static_cast<float*>(a.out0) + ((size_t)(img * a.H + yc) * a.W + x0) * a.Nout + nb32;
                        pstride = (size_t)a.Nout;
                        const float* act = static_cast<const float*>(a.act0);
                        if (act) {
                            float4 sl[4];
#pragma unroll
                            for (int q = 0; q < 4; ++q) sl[q] = *reinterpret_cast<const float4*>(act + pix * a.Nout + nbase + 8 * q);
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                v[q].x *= lrelu_slope(sl[q].x); v[q].y *= lrelu_slope(sl[q].y);
                                v[q].z *= lrelu_slope(sl[q].z); v[q].w *= lrelu_slope(sl[q].w);
                            }
                        }
                    }
                    f32_line_store(v, blk, pstride, lane, yok, a.W - x0);
                }
            }
        }
        if (t_next >= total_tiles) break;
        t = t_next;
    }
}

// ---- split-K finish: out = epilogue(sum over the K parts, in a fixed order) ---------------------------------------------------------
// One thread per (pixel or 2x2 pixel block, channel quad).  The epilogues are those of the conv kernels: EPI_FWD bias + LeakyReLU (+ the fused
// 2x2 max-pool when pool_out is set: the thread then owns a whole pooling window), EPI_GRAD LeakyReLU slope of the saved activation and the
// channel split of virtual concats.
template <bool POOL>
__global__ __launch_bounds__(256) void x3_splitk_finish_kernel(const ConvArgs a) {
    const int Q = a.Nout >> 2;
    const int Wb = POOL ? a.W >> 1 : a.W, Hb = POOL ? a.H >> 1 : a.H;
    const size_t total = (size_t)a.N * Hb * Wb * Q;
    const size_t plane = (size_t)a.N * a.H * a.W * a.Nout;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int q = (int)(i % Q);
        size_t p = i / Q;
        const int xb = (int)(p % Wb); p /= Wb;
        const int yb = (int)(p % Hb);
        const int img = (int)(p / Hb);
        const int n = 4 * q;
        float4 pooled = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int w = 0; w < (POOL ? 4 : 1); ++w) {
            const int y = POOL ? 2 * yb + (w >> 1) : yb, x = POOL ? 2 * xb + (w & 1) : xb;
            const size_t pix = (size_t)(img * a.H + y) * a.W + x;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int ks = 0; ks < a.ksplit; ++ks) {
                const float4 t = *reinterpret_cast<const float4*>(a.kpart + (size_t)ks * plane + pix * a.Nout + n);
                v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
            }
            if (a.epi == EPI_FWD) {
                const float4 b = *reinterpret_cast<const float4*>(a.bias + n);
                v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
                if (a.lrelu) { v.x = fmaxf(0.2f * v.x, v.x); v.y = fmaxf(0.2f * v.y, v.y); v.z = fmaxf(0.2f * v.z, v.z); v.w = fmaxf(0.2f * v.w, v.w); }
                *reinterpret_cast<float4*>(static_cast<float*>(a.out0) + pix * a.Nout + n) = v;
                if (POOL) pooled = w == 0 ? v : make_float4(fmaxf(pooled.x, v.x), fmaxf(pooled.y, v.y), fmaxf(pooled.z, v.z), fmaxf(pooled.w, v.w));
            } else {
                const bool lo = n < a.split;
                const int C = lo ? a.split : a.Nout - a.split;
                const size_t idx = pix * C + (lo ? n : n - a.split);
                const float* act = static_cast<const float*>(lo ? a.act0 : a.act1);
                if (act) {
                    const float4 s_ = *reinterpret_cast<const float4*>(act + idx);
                    v.x *= lrelu_slope(s_.x); v.y *= lrelu_slope(s_.y); v.z *= lrelu_slope(s_.z); v.w *= lrelu_slope(s_.w);
                }
                *reinterpret_cast<float4*>(static_cast<float*>(lo ? a.out0 : a.out1) + idx) = v;
            }
        }
        if (POOL) *reinterpret_cast<float4*>(static_cast<float*>(a.pool_out) + ((size_t)(img * Hb + yb) * Wb + xb) * a.Nout + n) = pooled;
    }
}

int launch_x3_splitk_finish(const ConvArgs& a, hipStream_t st) {
    const bool pool = a.epi == EPI_FWD && a.pool_out != nullptr;
    const size_t total = (size_t)a.N * (pool ? a.H / 2 : a.H) * (pool ? a.W / 2 : a.W) * (a.Nout / 4);
    if (!total) return 0;
    const unsigned grid = (unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    if (pool) { ELD_LAUNCH(x3_splitk_finish_kernel<true>, dim3(grid), dim3(256), 0, st, a); }
    else { ELD_LAUNCH(x3_splitk_finish_kernel<false>, dim3(grid), dim3(256), 0, st, a); }
    ELD_LAUNCH_CHECK();
    return 0;
}

template <int BN, int RPW, bool DB, bool BFIRST = false, bool BSLAB = false, bool STREAM = false>
int launch_x3(ConvArgs a, hipStream_t st) {
    a.xcd = eld_xcd_mask() & XCD_X3W;
    constexpr int TH = 4 * RPW;
    a.tiles_x = (a.W + TW - 1) / TW;
    a.tiles_y = (a.H + TH - 1) / TH;
    const size_t lds_bytes = (size_t)((TH + 2) * (TW + 2) + 3 * BN) * PX * sizeof(float);
    const long long tiles = (long long)a.tiles_x * a.tiles_y * a.N * (a.Nout / BN);
    if (tiles <= 0) return 0;
    if (tiles > 0x7fffffffLL) return ELD_ENOTSUP;
    auto kern = conv_x3_kernel<BN, RPW, DB, BFIRST, BSLAB, STREAM>;
    static EldAttrOnce once;
    { const int rc = once.ensure(kern, lds_bytes); if (rc) return rc; }
    int per_cu = (int)((160 * 1024) / lds_bytes);
    if (per_cu > 2) per_cu = 2;
    if (per_cu < 1 || (ELD_DBG(a) & 64)) per_cu = 1;
    long long grid = (long long)eld_num_cus() * per_cu;
    if (grid > tiles) grid = tiles;
    ELD_LAUNCH(kern, dim3((unsigned)grid), dim3(256), lds_bytes, st, a);
    ELD_LAUNCH_CHECK();
    return 0;
}

template <int BN, int RPW, int WAVES, bool DB, int STREAM = 0>
int launch_x3d(ConvArgs a, hipStream_t st) {
    a.xcd = eld_xcd_mask() & XCD_X3D;
    constexpr int TH = WAVES * RPW;
    conv_tile_shape(a.N, a.H, a.W, TH, a.pool_out != nullptr, a.tile_h, a.tile_w);
    a.vp = vrow_pitch(a.N, a.H, a.tile_h);
    a.tiles_x = (a.W + a.tile_w - 1) / a.tile_w;
    a.tiles_y = (vrow_extent(a.N, a.H, a.vp) + a.tile_h - 1) / a.tile_h;
    const size_t lds_bytes = (size_t)(TH + 2) * (TW + 2) * PX * sizeof(float) + 2 * (size_t)((3 * BN * PX * 4 + 1023) / 1024 * 1024);
    long long tiles = (long long)a.tiles_x * a.tiles_y * (a.Nout / BN);
    if (tiles <= 0) return 0;
    // split K where the smallest tiles still leave most of the chip idle (single patches: conv5_x of a 512 x 512 input is 32 workgroups) and the
    // caller provided room for the partial sums
    a.ksplit = 1;
    static const int no_splitk = [] { const char* e = getenv("ELD_NO_SPLITK"); return e ? atoi(e) : 0; }();      // dev switch (ELD_NO_SPLITK=1): small problems without the K split
    if (WAVES == 4 && a.kpart != nullptr && !no_splitk) {
        const int nch = (a.C0 + a.C1) / 16;
        const size_t plane = (size_t)a.N * a.H * a.W * a.Nout;
        int ks = (int)(eld_num_cus() / tiles);
        if (ks > nch / 2) ks = nch / 2;                       // at least two chunks per part
        if (ks > 16) ks = 16;
        while (ks > 1 && (size_t)ks * plane > a.kpart_floats) --ks;
        if (ks >= 2) { a.ksplit = ks; tiles *= ks; }
    }
    if (tiles > 0x7fffffffLL) return ELD_ENOTSUP;
    auto kern = conv_x3d_kernel<BN, RPW, WAVES, DB, STREAM>;
    static EldAttrOnce once;
    { const int rc = once.ensure(kern, lds_bytes); if (rc) return rc; }
    int per_cu = (int)((160 * 1024) / lds_bytes);
    if (per_cu > 2) per_cu = 2;
    if (per_cu < 1) per_cu = 1;
    long long grid = (long long)eld_num_cus() * per_cu;
    if (grid > tiles) grid = tiles;
    ELD_LAUNCH(kern, dim3((unsigned)grid), dim3(64 * WAVES), lds_bytes, st, a);
    ELD_LAUNCH_CHECK();
    // (Folding this finish into the last workgroup to deliver a part of a tile -- one arrival counter per tile behind a __threadfence() -- was built
    // and measured in round 4: the release fence writes the XCD's L2 back on every work item and conv_x3d_kernel<64,2,4> went from 36.6 to 88.3 us
    // per launch, 3.1 -> 4.3 ms per 512 x 512 step.  The kernel boundary is the cheap way to publish partial sums across XCDs.)
    if (a.ksplit > 1) return launch_x3_splitk_finish(a, st);
    return 0;
}

template <int MODE, int BN, int WAVES>
int launch_x3_gemm(ConvArgs a, hipStream_t st) {
    a.xcd = eld_xcd_mask() & XCD_GEMM;
    constexpr int TH = 2 * WAVES;
    a.tiles_x = (a.W + TW - 1) / TW;
    a.tiles_y = (a.H + TH - 1) / TH;
    const size_t lds_bytes = (size_t)(2 * TH * TW + 2 * BN) * PX * sizeof(float);
    const long long tiles = (long long)a.tiles_x * a.tiles_y * a.N * (a.Nout / BN);
    if (tiles <= 0) return 0;
    if (tiles > 0x7fffffffLL) return ELD_ENOTSUP;
    int per_cu = (int)((160 * 1024) / lds_bytes);
    if (per_cu > 2) per_cu = 2;
    long long grid = (long long)eld_num_cus() * per_cu;
    if (grid > tiles) grid = tiles;
    // (round 6: a software-pipelined variant -- 16-k stages, doubled stage buffer, the cut of stage s + 1 between the MFMAs of stage s -- measured +26 % per launch:
    // these launches are bound by the bytes a workgroup keeps in flight, not by the cut; commit 7f24c6e, profiles/r06_ab_notes.md section 9)
    auto kern = conv_x3_gemm_kernel<MODE, BN, WAVES>;
    static EldAttrOnce once;
    { const int rc = once.ensure(kern, lds_bytes); if (rc) return rc; }
    ELD_LAUNCH(kern, dim3((unsigned)grid), dim3(64 * WAVES), lds_bytes, st, a);
    ELD_LAUNCH_CHECK();
    return 0;
}

}  // namespace


static bool x3_32_slabs() {
    static const int on = [] { const char* e = getenv("ELD_X3_BSLAB"); return e ? atoi(e) : 0; }();      // measured 0.0 ... +0.3 % against the per-stage cut (profiles/r06_ab_notes.md): opt-in
    return on != 0 || x3w_enabled();             // conv_x3w_kernel takes slabs: then every 32-output-channel layer is packed that way (small problems: the BSLAB variant)
}
static bool x3d_32_kernel(int N, int H, int W) {      // opt-in ELD_X3D_32=1: the 32-channel layers on the LDS-DMA kernel (measured slower, see x3_slab_bn)
    static const int on = [] { const char* e = getenv("ELD_X3D_32"); return e ? atoi(e) : 0; }();
    return on && x3_32_slabs() && conv_tile_count(N, H, W, 32, false) >= 2 * eld_num_cus();
}

// Tile shapes of the LDS-DMA kernel (measured, profiles/r02_*): 16-row tiles with ONE 8-wave workgroup per CU; 128 output channels per
// tile where the layer has them (half the halo cuts per MFMA), else 64.  Layers with 32 output channels stay on the register-staged
// conv_x3_kernel<32, 4> with two workgroups per CU: their K loop is 2-4 chunks long and a lone workgroup cannot hide its epilogue
// (DMA variants with 32-channel slabs measured 0-2 % slower).
// Small problems (the reference trains on single 512 x 512 patches, train_syn.py defaults): when the big tiles do not give every CU a
// workgroup the layer drops to 64-channel tiles, and then to 8-row tiles with 4-wave workgroups -- 4x the workgroups of the 16 x 128 tile
// (conv5_x of a 512 x 512 patch: 8 -> 32 workgroups).  The pack kernel lays the slabs out for the same choice (same function, same arguments).
int x3_slab_bn(int Nout, int N, int H, int W, int* waves) {
    if (waves) *waves = 8;
    // 32 output channels (round 4, opt-in ELD_X3D_32=1): conv_x3d_kernel<32, 4, 8> -- 32-row x 32-pixel tiles, the layer's pre-split weight slabs by
    // LDS-DMA instead of a cut of the fp32 weights per stage and tile (a third of conv_x3_kernel<32, 4>'s staging work).  Measured 6 % SLOWER than
    // conv_x3_kernel<32, 4> (3087 vs 2917 us per launch, same box): one 8-wave workgroup per CU loses more to its barriers than the two 4-wave
    // workgroups of the register-staged kernel lose to the weight cut.  Off by default.
    if (Nout == 32) {
        // (round 5: conv_x3d_kernel<32, 2, 4> -- the same slabs, 8-row tiles, two 4-wave workgroups per CU -- measured +10.7 %: not kept either)
        // Round 6 (opt-in ELD_X3_BSLAB=1): the 32-channel layers take pre-split slabs too -- consumed through registers by conv_x3_kernel<32, 4, ..., BSLAB>:
        // a fifth of the kernel's VALU instructions gone and no change in its time (2848-2862 vs 2842-2854 us per launch, same box, interleaved): the
        // kernel is not bound by VALU issue.  Default: fp32 packed weights cut per stage, the round-5 kernel.
        return x3_32_slabs() ? 32 : 0;
    }
    if (Nout % 64) return 0;
    const long long px_tiles = conv_tile_count(N, H, W, 16, false);
    const int cus = eld_num_cus();
    if (Nout % 128 == 0 && px_tiles * (Nout / 128) >= cus) return 128;
    if (waves && px_tiles * (Nout / 64) < cus) *waves = 4;
    // (round 5: the 4-wave kernel for the 64-channel layers of big problems too -- 8-row tiles, two 81 KB workgroups per CU -- measured -0.2 %: not kept)
    return 64;
}

// XCD-aware work-item ids (conv.h xcd_block): one bit per kernel family, env ELD_XCD for same-box A/B runs
int eld_xcd_mask() {
    static const int m = [] { const char* e = getenv("ELD_XCD"); return e ? atoi(e) : 127; }();
    return m;
}

int eld_tile_band() {
    static const int b = [] { const char* e = getenv("ELD_TILE_BAND"); return e ? atoi(e) : 4; }();
    return b;
}

// a: fp32 CONV_3X3 arguments already validated by launch_conv
int launch_conv_x3(const ConvArgs& a_in, hipStream_t st) {
    ConvArgs a = a_in;
    a.prof = nullptr;
    if ((size_t)a.H * a.W * a.C0 * 4 >= 0xFFFFFFF0ull) return ELD_ENOTSUP;
    if (a.pool_out && (a.epi != EPI_FWD || (a.H & 1) || (a.W & 1))) return ELD_EINVAL;
    int waves = 8;
    const int bn0 = x3_slab_bn(a.Nout, a.N, a.H, a.W, nullptr);
    // conv_x3d_kernel addresses a two-image window (virtual rows)
    if ((bn0 >= 64 || (bn0 == 32 && x3d_32_kernel(a.N, a.H, a.W))) && a.N > 1 && (size_t)a.H * a.W * a.C0 * 8 >= 0xFFFFFFF0ull) return ELD_ENOTSUP;
    const int bn = x3_slab_bn(a.Nout, a.N, a.H, a.W, &waves);
    // round 6: streamed fragment blocks (x3_stage_blocks) in the main loops; ELD_X3_STREAM=0 restores hipcc's own schedule of the round-5 loops (A/B runs),
    // a bit mask selects per family: 1 = the 32-channel kernel, 2 = conv_x3d_kernel<64>, 4 = conv_x3d_kernel<128>
    static const int stream = [] { const char* e = getenv("ELD_X3_STREAM"); return e ? atoi(e) : 7; }();
    if (bn == 32 && x3w_takes(a)) return launch_conv_x3w(a, st);      // specialised waves (conv_x3w.hip)
    if (bn == 32 && x3d_32_kernel(a.N, a.H, a.W)) return launch_x3d<32, 4, 8, false>(a, st);
    if (bn == 32) return (stream & 1) ? launch_x3<32, 4, false, true, true, true>(a, st) : launch_x3<32, 4, false, true, true>(a, st);      // pre-split slabs through registers (round 6)
    // ELD_X3_WREUSE (bit mask: 2 = conv_x3d_kernel<64>, 4 = conv_x3d_kernel<128>; default both): the streamed blocks in the weight-reuse order (conv_x3_dev.h
    // x3_stage_blocks) -- a third fewer LDS fragment reads; same box, interleaved: 2168 -> 2126 us (-1.9 %) / 2580 -> 2538 us (-1.6 %) per launch, step -0.9 %
    static const int wreuse = [] { const char* e = getenv("ELD_X3_WREUSE"); return e ? atoi(e) : 6; }();
    if (bn == 128) {                                                  // weights pre-split in slab layout: LDS-DMA kernel
        if ((stream & 4) && (wreuse & 4)) return launch_x3d<128, 2, 8, false, 2>(a, st);
        return (stream & 4) ? launch_x3d<128, 2, 8, false, 1>(a, st) : launch_x3d<128, 2, 8, false>(a, st);
    }
    if (bn == 64) {
        if (waves != 8) return launch_x3d<64, 2, 4, false>(a, st);
        if ((stream & 2) && (wreuse & 2)) return launch_x3d<64, 2, 8, false, 2>(a, st);
        return (stream & 2) ? launch_x3d<64, 2, 8, false, 1>(a, st) : launch_x3d<64, 2, 8, false>(a, st);
    }
    // round 5: the next stage's slab loads are issued AHEAD of the halo loads (template BFIRST; -2.7 % per launch, same box); ELD_X3_BFIRST=0 restores the
    // round-4 order for A/B runs
    static const int bfirst = [] { const char* e = getenv("ELD_X3_BFIRST"); return e ? atoi(e) : 1; }();
    if (bfirst) return (stream & 1) ? launch_x3<32, 4, false, true, false, true>(a, st) : launch_x3<32, 4, false, true>(a, st);
    return launch_x3<32, 4, false>(a, st);
}

// transposed-conv directions (CONV_1X1 + EPI_CONVT_FWD, CONV_GATHER2X2 + EPI_GRAD); returns ELD_ENOTSUP for shapes the GEMM
// tiling does not cover (the caller then uses the fp32-MFMA kernel)
int launch_conv_x3_gemm(const ConvArgs& a, int mode, hipStream_t st) {
    const size_t src_img = (size_t)a.H * a.W * a.C0 * 4 * (mode == CONV_GATHER2X2 ? 4 : 1);
    if (a.Nout % 64 || a.C0 % 32 || a.C1 != 0 || src_img >= 0xFFFFFFF0ull) return ELD_ENOTSUP;
    // (an 8-wave 16 x 32 x 128 tile, <.., 128, 8>, was measured at the same speed as the 4-wave 8 x 32 x 64 one: these launches are not bound by
    // the operand cuts; the small tile wastes less on the 89 x 133 / 178 x 266 levels)
    if (mode == CONV_1X1 && a.epi == EPI_CONVT_FWD) return launch_x3_gemm<CONV_1X1, 64, 4>(a, st);
    if (mode == CONV_GATHER2X2 && a.epi == EPI_GRAD && a.split == a.Nout) return launch_x3_gemm<CONV_GATHER2X2, 64, 4>(a, st);
    return ELD_ENOTSUP;
}
