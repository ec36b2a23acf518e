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
#include <stdlib.h>
#include "conv.h"

#define TW 32
#define JB 32     // j (shifted-operand channel) tile

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef int i32x4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

// BFM = true (bf16 inputs only): the tiles stay bf16 in LDS, untransposed [pixel][channel], and the contraction runs on
// v_mfma_f32_32x32x16_bf16 with K = 16 consecutive pixels of a row: lane (channel = lane&31, kb = lane>>5) gathers its
// 8 pixel values with 16-bit LDS reads (a half-wave reads 32 consecutive channels of one pixel: conflict-free).  The
// G fragment of a k-step is shared by all taps, so a k-step costs 8 + 8*TAPS ds_read_u16 for TAPS MFMAs.
//
// ALG_X3 (fp32 inputs): as conv_x3.hip -- every fp32 value is cut exactly into three bf16 pieces while it is staged
// (three [pixel][channel] bf16 planes per tile in LDS) and each 16-pixel k-step accumulates the six piece products
// g1x1 + g1x2 + g2x1 + g1x3 + g2x2 + g3x1 on v_mfma_f32_32x32x16_bf16: fp32-level accuracy at 6 x 32 matrix-pipe
// cycles per 16 k instead of 8 x 64.  The three horizontal taps of a kernel row share one 10-pixel window per piece
// (10 ds_read_u16 + 4 v_alignbit instead of 24 reads).
enum { ALG_F32 = 0, ALG_BFM = 1, ALG_X3 = 2, ALG_H2 = 3 };      // ALG_H2: two fp16 pieces per fp32 operand, three products (conv_fp32_algo 2)

template <typename T, int MODE, int WCO, int TH, int ALG>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(const WgradArgs a) {
    constexpr bool BFM = ALG == ALG_BFM, X3 = ALG == ALG_X3, H2 = ALG == ALG_H2;
    static_assert(!(X3 || H2) || sizeof(T) == 4, "x3 / h2: fp32 inputs");
    constexpr int ES = sizeof(T);            // element size of g / x in HBM (float or bf16); accumulation is fp32
    constexpr int EPU = 16 / ES;             // elements per 16-byte staging unit
    constexpr int LES = X3 ? 6 : (BFM ? 2 : 4);   // bytes per element in LDS (x3: three bf16 planes; h2: two fp16 planes = 4)
    constexpr int TAPS = MODE == CONV_3X3 ? 9 : 4;
    constexpr int WPIX = 4 / WCO;
    constexpr int COB = 32 * WCO;
    constexpr int TPIX = TH * TW;
    constexpr int PW = TPIX / WPIX;        // pixels per wave per tile
    constexpr int KS = PW / 2;             // MFMA k-steps per wave per tile
    constexpr int X_PIX = MODE == CONV_3X3 ? (TH + 2) * (TW + 2) : 4 * TPIX;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // piece-plane layouts of fp32 inputs (ALG_X3 / ALG_H2): a 32-channel block of G is padded by one 64-byte row, as in wgrad8_kernel -- the staging
    // stores of one pixel's two blocks (consecutive lanes) then land on different banks (block planes of TPIX x 64 B are multiples of the 256-byte
    // bank row: 2-way conflicts on every G store without the pad; round 4's counters: SQ_LDS_BANK_CONFLICT = the kernel's MFMA count)
    constexpr int GROWS = TPIX + ((X3 || H2) ? 1 : 0);   // rows of a 32-channel block of G in LDS
    float* ldsG = lds;                                   // [TPIX][COB]   (LES-byte elements); piece layouts: [plane][COB / 32][GROWS][32]
    float* ldsX = lds + GROWS * COB * LES / 4;           // [X_PIX][JB]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = lane & 31, hi = lane >> 5;
    const int wco = wave % WCO, wpix = wave / WCO;
    const int IB = a.CA / COB, JBn = a.CBp / JB;
    int bid = xcd_block(a.xcd);
    const int jb = bid % JBn; bid /= JBn;
    const int ib = bid % IB;
    const int ps = bid / IB;
    const int i0 = ib * COB, j0 = jb * JB;
    const int CB = a.C0 + a.C1;
    const int Hx = MODE == CONV_GATHER2X2 ? 2 * a.H : a.H;
    const int Wx = MODE == CONV_GATHER2X2 ? 2 * a.W : a.W;
    // source of the j tile (virtual concat); channels >= CB are zero padding
    const char* xsrc; int Cs, cs;
    if (j0 < a.C0) { xsrc = static_cast<const char*>(a.x0); Cs = a.C0; cs = j0; } else { xsrc = static_cast<const char*>(a.x1); Cs = a.C1; cs = j0 - a.C0; }
    const int jvalid = min(JB, CB - j0 > 0 ? (j0 < a.C0 ? a.C0 - j0 : CB - j0) : 0);   // channels of this tile that exist

    float sc_g = 1.0f, sc_x = 1.0f;              // H2: power-of-two operand scales
    if constexpr (H2) {
        sc_g = h2_scale(*a.amax_g);
        float mx = *a.amax_x0;
        if (a.amax_x1) mx = fmaxf(mx, *a.amax_x1);
        sc_x = h2_scale(mx);
    }
    f32x16 acc[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    float bsum = 0.f;
    // column sums of the gathered operand (transposed-conv bias gradient): a thread always stages the same channel group (256 % (JB/EPU) == 0)
    constexpr bool XSUM = MODE == CONV_GATHER2X2;
    const bool xsum_on = XSUM && a.xbpart != nullptr && ib == 0;
    float xs[EPU];                           // 4 channels (fp32 units) or 8 (bf16 units) of this thread's quad / octet
#pragma unroll
    for (int e = 0; e < EPU; ++e) xs[e] = 0.f;

    const int tiles_per_img = a.tiles_x * a.tiles_y;
    const int ntiles = tiles_per_img * a.N;
    // Register-staged pipeline: tile t+1's global loads (buffer loads: 32-bit offsets, SGPR descriptor, hardware
    // zero-fill for the out-of-image marker) are in flight during tile t's MFMA phase and go to LDS after the barrier.
    constexpr int G_UNITS = TPIX * (COB / EPU), X_UNITS = X_PIX * (JB / EPU);
    constexpr int G_IT = (G_UNITS + 255) / 256, X_IT = (X_UNITS + 255) / 256;
    constexpr unsigned OOB = 0xFFFFFFF0u;
    float4 rg[G_IT], rx[X_IT];               // raw 16-byte units (4 fp32 or 8 bf16)
    const size_t g_img = (size_t)a.H * a.W * a.CA * ES, x_img = (size_t)Hx * Wx * Cs * ES;     // bytes
    auto load_tile = [&](int tile) {
        const int img = tile / tiles_per_img;
        const int trem = tile - img * tiles_per_img;
        const int ty = trem / a.tiles_x, tx = trem - ty * a.tiles_x;
        const int y0 = ty * TH, x0 = tx * TW;
        const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc((void*)(static_cast<const char*>(a.g) + (size_t)img * g_img), 0, (int)g_img, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)(xsrc + (size_t)img * x_img), 0, (int)x_img, 0x00020000);
        unsigned og[G_IT], ox[X_IT];
#pragma unroll
        for (int it = 0; it < G_IT; ++it) {
            const int u = tid + it * 256;
            const int lp = u / (COB / EPU), part = u - lp * (COB / EPU);
            const int py = lp / TW, px = lp - py * TW;
            const int gy = y0 + py, gx = x0 + px;
            og[it] = (u < G_UNITS && gy < a.H && gx < a.W) ? (unsigned)((gy * a.W + gx) * a.CA * ES + part * 16) : OOB;
        }
#pragma unroll
        for (int it = 0; it < X_IT; ++it) {
            const int u = tid + it * 256;
            const int hp = u / (JB / EPU), part = u - hp * (JB / EPU);
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
            ox[it] = (u < X_UNITS && ok && part * EPU < jvalid) ? (unsigned)((gy * Wx + gx) * Cs * ES + part * 16) : OOB;
        }
#pragma unroll
        for (int it = 0; it < G_IT; ++it) rg[it] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs_g, (int)og[it], i0 * ES, 0));
#pragma unroll
        for (int it = 0; it < X_IT; ++it) rx[it] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)ox[it], cs * ES, 0));
    };
    constexpr int GPL = GROWS * COB, XPL = X_PIX * JB;                // x3: elements per piece plane
    auto put = [&](float* dst, int u, const float4& raw, int plane_elems, int e0) {      // e0: element index of the unit inside a bf16 plane            // LDS rows are unit-linear: [lp][COB] and [hp][JB]
        if constexpr (H2) {
            typedef __fp16 h2v __attribute__((ext_vector_type(2)));
            bf16_t* d = reinterpret_cast<bf16_t*>(dst) + e0;
            const float sc = plane_elems == GPL ? sc_g : sc_x;
            const float x0 = raw.x * sc, x1 = raw.y * sc, x2 = raw.z * sc, x3 = raw.w * sc;
            const h2v a01 = __builtin_amdgcn_cvt_pkrtz(x0, x1), a23 = __builtin_amdgcn_cvt_pkrtz(x2, x3);
            const h2v b01 = {(__fp16)(x0 - (float)a01[0]), (__fp16)(x1 - (float)a01[1])}, b23 = {(__fp16)(x2 - (float)a23[0]), (__fp16)(x3 - (float)a23[1])};      // rounded to nearest
            *reinterpret_cast<uint2*>(d) = make_uint2(__builtin_bit_cast(unsigned, a01), __builtin_bit_cast(unsigned, a23));
            *reinterpret_cast<uint2*>(d + plane_elems) = make_uint2(__builtin_bit_cast(unsigned, b01), __builtin_bit_cast(unsigned, b23));
        } else if constexpr (X3) {
            bf16_t* d = reinterpret_cast<bf16_t*>(dst) + e0;
            const unsigned x0 = __float_as_uint(raw.x), x1 = __float_as_uint(raw.y), x2 = __float_as_uint(raw.z), x3 = __float_as_uint(raw.w);
            const float r0 = raw.x - __uint_as_float(x0 & 0xFFFF0000u), r1 = raw.y - __uint_as_float(x1 & 0xFFFF0000u);
            const float r2 = raw.z - __uint_as_float(x2 & 0xFFFF0000u), r3 = raw.w - __uint_as_float(x3 & 0xFFFF0000u);
            const unsigned y0 = __float_as_uint(r0), y1 = __float_as_uint(r1), y2 = __float_as_uint(r2), y3 = __float_as_uint(r3);
            const unsigned z0 = __float_as_uint(r0 - __uint_as_float(y0 & 0xFFFF0000u)), z1 = __float_as_uint(r1 - __uint_as_float(y1 & 0xFFFF0000u));
            const unsigned z2 = __float_as_uint(r2 - __uint_as_float(y2 & 0xFFFF0000u)), z3 = __float_as_uint(r3 - __uint_as_float(y3 & 0xFFFF0000u));
            auto hp = [](unsigned lo, unsigned hi_) { return __builtin_amdgcn_perm(hi_, lo, 0x07060302u); };
            *reinterpret_cast<uint2*>(d) = make_uint2(hp(x0, x1), hp(x2, x3));
            *reinterpret_cast<uint2*>(d + plane_elems) = make_uint2(hp(y0, y1), hp(y2, y3));
            *reinterpret_cast<uint2*>(d + 2 * plane_elems) = make_uint2(hp(z0, z1), hp(z2, z3));
        } else if constexpr (BFM) {
            *reinterpret_cast<float4*>(reinterpret_cast<bf16_t*>(dst) + e0) = raw;      // 8 bf16 as loaded
        } else if constexpr (ES == 4) {
            *reinterpret_cast<float4*>(dst + u * 4) = raw;            // 16 bytes as loaded
        } else {                                                      // 8 bf16 -> 8 fp32
            const uint4 q = __builtin_bit_cast(uint4, raw);
            *reinterpret_cast<float4*>(dst + u * 8) = unpack_bf4(make_uint2(q.x, q.y));
            *reinterpret_cast<float4*>(dst + u * 8 + 4) = unpack_bf4(make_uint2(q.z, q.w));
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int it = 0; it < G_IT; ++it) {
            const int u = tid + it * 256;
            // bf16 planes of G are [32-channel block][pixel][32]: 64-byte rows, the layout ds_read_b64_tr_b16 reads conflict-free
            const int lp = u / (COB / EPU), part = u - lp * (COB / EPU);
            const int c = part * EPU;
            if (u < G_UNITS) put(ldsG, u, rg[it], GPL, ((c >> 5) * GROWS + lp) * 32 + (c & 31));
        }
#pragma unroll
        for (int it = 0; it < X_IT; ++it) {
            const int u = tid + it * 256;
            if (u < X_UNITS) put(ldsX, u, rx[it], XPL, u * EPU);
        }
        if constexpr (XSUM) {
            if (xsum_on) {
#pragma unroll
                for (int it = 0; it < X_IT; ++it) {                            // out-of-image units load zeros
                    if constexpr (ES == 4) { xs[0] += rx[it].x; xs[1] += rx[it].y; xs[2] += rx[it].z; xs[3] += rx[it].w; }
                    else {
                        const uint4 q = __builtin_bit_cast(uint4, rx[it]);
                        const float4 lo = unpack_bf4(make_uint2(q.x, q.y)), hi4 = unpack_bf4(make_uint2(q.z, q.w));
                        xs[0] += lo.x; xs[1] += lo.y; xs[2] += lo.z; xs[3] += lo.w; xs[4] += hi4.x; xs[5] += hi4.y; xs[6] += hi4.z; xs[7] += hi4.w;
                    }
                }
            }
        }
    };

    // LDS read bases of this lane: step s of the K loop reads pixel lp0 + s (no row wrap inside a wave's slice),
    // so every read is base + compile-time offset.
    const int lp0 = wpix * PW + hi * KS;
    const int py0 = lp0 / TW, px0 = lp0 - py0 * TW;
    const float* gbase = ldsG + lp0 * COB + wco * 32 + m;
    const float* xbase = ldsX + (MODE == CONV_3X3 ? (py0 * (TW + 2) + px0) : lp0) * JB + m;

    if (ps < ntiles) load_tile(ps);
    for (int tile = ps; tile < ntiles; tile += a.psplit) {
        __syncthreads();
        store_tile();
        __syncthreads();
        if (tile + a.psplit < ntiles) load_tile(tile + a.psplit);
        if constexpr (X3 || BFM || H2) {
        // ---- bf16 / fp16 MFMA, K = 16 consecutive pixels of a row per k-step.  Operands come straight out of the untransposed
        //      [pixel][32 channels] bf16 planes with ds_read_b64_tr_b16: a 16-lane group reads a [4 pixels][16 channels]
        //      block (lane i supplies the 8 bytes at pixel i/4, channels 4(i%4)..+3) and lane i receives the 4 pixel values
        //      of channel i -- two reads make one 8-k MFMA operand, no 16-bit gathers, no packing.  Shifted taps are just
        //      other immediate offsets (rows are 64 bytes, any pixel offset keeps the 8-byte alignment).
        constexpr int NPC = X3 ? 3 : (H2 ? 2 : 1);
        constexpr int KSB = PW / 16;
        const int gi = lane & 15, gg = lane >> 4;                   // lane inside its 16-lane group, group: channels 16(gg&1).., k-half gg>>1 (= hi)
    const int lq0 = wpix * PW + 8 * hi + (gi >> 2);             // pixel row this lane ADDRESSES in k-step 0 (first 4-pixel block)
        const int pyq = lq0 / TW, pxq = lq0 - pyq * TW;
        const bf16_t* gq = reinterpret_cast<const bf16_t*>(ldsG) + (wco * GROWS + lq0) * 32 + (gg & 1) * 16 + (gi & 3) * 4;
        const bf16_t* xq = reinterpret_cast<const bf16_t*>(ldsX) + (MODE == CONV_3X3 ? (pyq * (TW + 2) + pxq) : lq0) * JB + (gg & 1) * 16 + (gi & 3) * 4;
        auto tr8 = [](const bf16_t* p0) {                           // pixels [0,4) and [4,8) of the lane's k-half -> one operand
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p0);
            const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p0 + 4 * 32));
            return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi4, 0, 1, 2, 3, 4, 5, 6, 7));
        };
#pragma unroll
        for (int ks = 0; ks < KSB; ++ks) {
            const int lrel = ks * 16;
            const int dy0 = lrel / TW, dxp = lrel - dy0 * TW;
            bf16x8 ga[NPC];
#pragma unroll
            for (int pc = 0; pc < NPC; ++pc) {
                ga[pc] = tr8(gq + pc * GPL + lrel * 32);
                if (a.bpart && H2) {
                    const half8 hq = __builtin_bit_cast(half8, ga[pc]);
#pragma unroll
                    for (int e = 0; e < 8; ++e) bsum += (float)hq[e];
                } else if (a.bpart) {
                    const uint4 q = __builtin_bit_cast(uint4, ga[pc]);
                    bsum += __uint_as_float(q.x << 16) + __uint_as_float(q.x & 0xFFFF0000u) + __uint_as_float(q.y << 16) + __uint_as_float(q.y & 0xFFFF0000u)
                          + __uint_as_float(q.z << 16) + __uint_as_float(q.z & 0xFFFF0000u) + __uint_as_float(q.w << 16) + __uint_as_float(q.w & 0xFFFF0000u);
                }
            }
            constexpr int TROW = MODE == CONV_3X3 ? 3 : TAPS;       // taps handled together (a kernel row / all four gather taps)
#pragma unroll
            for (int t0 = 0; t0 < TAPS; t0 += TROW) {
                bf16x8 xb[TROW][NPC];
#pragma unroll
                for (int tt = 0; tt < TROW; ++tt) {
                    const int t = t0 + tt;
                    const int xoff = MODE == CONV_3X3 ? ((dy0 + t / 3) * (TW + 2) + dxp + t % 3) : (t * TPIX + lrel);
#pragma unroll
                    for (int pc = 0; pc < NPC; ++pc) xb[tt][pc] = tr8(xq + pc * XPL + xoff * JB);
                }
                if constexpr (H2) {
                    constexpr int GI[3] = {0, 1, 0};
                    constexpr int XI[3] = {1, 0, 0};
#pragma unroll
                    for (int q = 0; q < 3; ++q)
#pragma unroll
                        for (int tt = 0; tt < TROW; ++tt)
                            acc[t0 + tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, ga[GI[q]]), __builtin_bit_cast(half8, xb[tt][XI[q]]),
                                                                                  acc[t0 + tt], 0, 0, 0);
                } else if constexpr (X3) {
                    constexpr int GI[6] = {0, 1, 2, 0, 1, 0};
                    constexpr int XI[6] = {2, 1, 0, 1, 0, 0};
#pragma unroll
                    for (int q = 0; q < 6; ++q)
#pragma unroll
                        for (int tt = 0; tt < TROW; ++tt)
                            acc[t0 + tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga[GI[q]], xb[tt][XI[q]], acc[t0 + tt], 0, 0, 0);
                } else {
#pragma unroll
                    for (int tt = 0; tt < TROW; ++tt) acc[t0 + tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga[0], xb[tt][0], acc[t0 + tt], 0, 0, 0);
                }
                if constexpr ((X3 || H2) && MODE == CONV_3X3) __builtin_amdgcn_sched_barrier(0);      // keep the next row's reads below: 256-VGPR budget
            }
        }
        } else if constexpr (!BFM) {
        // ---- K loop, fully unrolled; operands of step s+1 are read from LDS under step s's MFMAs -------------
        float fa[2], fb[2][TAPS];
        auto read_step = [&](int s_, float& A, float (&B)[TAPS]) {
            A = gbase[s_ * COB];
#pragma unroll
            for (int t = 0; t < TAPS; ++t) {
                const int toff = MODE == CONV_3X3 ? ((t / 3) * (TW + 2) + t % 3) : t * TPIX;
                B[t] = xbase[(toff + s_) * JB];
            }
        };
        read_step(0, fa[0], fb[0]);
#pragma unroll
        for (int s_ = 0; s_ < KS; ++s_) {
            const int cur = s_ & 1;
            if (s_ + 1 < KS) read_step(s_ + 1, fa[cur ^ 1], fb[cur ^ 1]);
            __builtin_amdgcn_sched_barrier(0);
            bsum += fa[cur];
#pragma unroll
            for (int t = 0; t < TAPS; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur], fb[cur][t], acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
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
                a.part[pbase + ((size_t)t * a.CA + i0 + wco * 32 + row) * a.CBp + j0 + m] = H2 ? acc[t][i] * (1.0f / sc_g) * (1.0f / sc_x) : acc[t][i];
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
            a.bpart[(size_t)ps * a.CA + i0 + tid] = H2 ? s * (1.0f / sc_g) : s;
        }
    }
    if constexpr (XSUM) {
        if (xsum_on) {                                  // block-uniform: 32 threads hold each channel quad; fixed-order combine through LDS
            __syncthreads();
            float* r = lds;                             // [256 threads][EPU]
#pragma unroll
            for (int e = 0; e < EPU; ++e) r[tid * EPU + e] = xs[e];
            __syncthreads();
            if (tid < JB) {
                constexpr int PARTS = JB / EPU;         // threads tid, tid + PARTS, ... stage the same channel group
                const int part = tid / EPU, comp = tid % EPU;
                float s = 0.f;
                for (int k = 0; k < 256 / PARTS; ++k) s += r[(k * PARTS + part) * EPU + comp];
                a.xbpart[(size_t)ps * a.CBp + j0 + tid] = s;
            }
        }
    }
}

// =============================================================================================================================
// wgrad8_kernel -- the three-piece (ALG_X3) 3x3 weight gradient re-blocked for ONE 8-wave workgroup per CU:
//   output block = (32*WCO out-channels) x (32*WCI in-channels) x 9 taps, waves = WCO x WCI x WPIX (pixel slices); a wave still owns
//   one 32x32 tile of all 9 taps (144 accumulator registers) and walks the spatial tiles assigned to its workgroup.
// Against wgrad_kernel<float, CONV_3X3, .., ALG_X3> (64 x 32 blocks, 4 waves, two workgroups per CU):
//   * a staged G / X tile feeds 2-4x more MFMAs (G is cut into pieces once per 128 in-channel... once per block column, X once per
//     block row): split VALU, staging loads and LDS writes per MFMA drop accordingly;
//   * bias sums come from the staging registers (a thread always stages the same channel quad), not from unpacked fragments;
//   * tile offsets are a scalar tile base + per-thread constants (no per-tile division chains);
//   * small blocks take taller tiles (TH = 4 / 8 rows): the 3x3 halo overhead of X falls from 2.1x to 1.6x / 1.3x.
// LDS planes are [32-channel block][pixel][32] bf16 (64-byte rows: what ds_read_b64_tr_b16 reads conflict-free), three planes
// (pieces) per operand.  Partials / reduction kernel / numerics are those of wgrad_kernel.
// Round 4: the spatial tile is TH x TWT pixels with TWT in {8, 16, 32} (a 16-pixel k-step = two rows of 8, one row of 16 or half a row of 32: the
// fragment offsets below stay compile-time for all three) over the VIRTUAL-ROW strip of the batch (conv.h vrow_*).  Pixels are the K dimension
// of this GEMM, so every padding pixel of a ragged level is a wasted MFMA column: 8-wide tiles bring 89 x 133 from 1.30x (2 x 32 tiles) to 1.03x,
// and their 3x3 halo is smaller too (10 x 10 against 4 x 34 pixels of X per 64 pixels of G).
template <typename T, int WCO, int WCI, int WPIX, int TH, int TWT, int STREAM = 0>      // T = float (three bf16 pieces per operand, six products) or bf16_t (the tiles as they are, one product); STREAM: 1 = streamed blocks, 2 = row-shared X fragments (below)
__global__ __launch_bounds__(512, 2) void wgrad8_kernel(const WgradArgs a) {
    constexpr int ES = sizeof(T), EPU = 16 / ES, NPC = ES == 4 ? 3 : 1, XQ = 32 / EPU;      // element size, elements per 16-byte unit, pieces, units per 32-channel pixel
    static_assert(WCO * WCI * WPIX == 8, "8 waves");
    constexpr int THREADS = 512, TAPS = 9;
    constexpr int COB = 32 * WCO, JBK = 32 * WCI, TPIX = TH * TWT, PW = TPIX / WPIX, KSB = PW / 16;
    static_assert(TWT == 8 || TWT == 16 || TWT == 32, "tile width");
    static_assert(PW % 16 == 0 && PW >= 16, "a wave's pixel slice is a whole number of 16-pixel k-steps");
    constexpr int X_PIX = (TH + 2) * (TWT + 2);
    // a 32-channel block of G is padded by one 64-byte row: the staging stores of one pixel's blocks (consecutive lanes) land on different banks
    // (block planes of TPIX x 64 B are multiples of the 256-byte bank row: 2-way conflicts on every G store without the pad)
    constexpr int GBLK = TPIX * 32 + (ES == 4 ? 32 : 0);      // (fp32 inputs only: -6 % on the 128 x 64 blocks; the bf16 kernels measured +2...+3 % with the pad)
    constexpr int GPL = GBLK * WCO, XPL = X_PIX * JBK;                 // elements per piece plane
    constexpr int G_Q = COB / EPU, G_UNITS = TPIX * G_Q, G_IT = G_UNITS / THREADS;
    static_assert(G_UNITS % THREADS == 0 && THREADS % G_Q == 0, "a thread stages the same channel quad of G in every iteration");
    constexpr int X_UNITS1 = X_PIX * XQ, X_IT = (X_UNITS1 + THREADS - 1) / THREADS;     // per 32-channel block of X
    extern __shared__ __attribute__((aligned(16))) float lds[];
    bf16_t* ldsG = reinterpret_cast<bf16_t*>(lds);                    // [3][WCO][TPIX][32]
    bf16_t* ldsX = ldsG + NPC * GPL;                                  // [NPC][WCI][X_PIX][32]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = lane & 31, hi = lane >> 5;
    const int wco = wave % WCO, wci = (wave / WCO) % WCI, wpix = wave / (WCO * WCI);
    const int IB = a.CA / COB, JBn = a.CBp / JBK;
    int bid = xcd_block(a.xcd);
    const int jb = bid % JBn; bid /= JBn;
    const int ib = bid % IB;
    const int ps = bid / IB;
    const int i0 = ib * COB, j0 = jb * JBK;
    const bool do_bias = a.bpart != nullptr && jb == 0;

    f32x16 acc[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    float4 bsum4 = make_float4(0.f, 0.f, 0.f, 0.f), bsum4b = make_float4(0.f, 0.f, 0.f, 0.f);      // bias sums of this thread's channel group (second quad: bf16 units of 8)

    const int ntiles = a.tiles_x * a.tiles_y;                         // tiles_y counts TH-row tiles of the virtual-row strip (all images)
    const int VP = a.vp;
    const bool seam = VP % TH != 0;                                   // tiles may straddle two images (conv.h vrow_pitch)
    constexpr unsigned OOB = 0xFFFFFFF0u;
    // ---- per-thread staging constants -----------------------------------------------------------------------------------------
    // G unit u = tid + it*512: pixel lp = u / G_Q, channel quad part = u % G_Q (the same for every `it`)
    //   -> pixel lp0 + it*G_STEP with lp0 = tid / G_Q < G_STEP: row / column of every iteration follow from (lp0, it) by constants
    constexpr int G_STEP = THREADS / G_Q;                             // 16, 32 or 64 pixels per pass
    const int g_part = tid % G_Q, g_lp0 = tid / G_Q;
    const int g_px0 = g_lp0 % TWT, g_py0 = g_lp0 / TWT;
    const unsigned g_off0 = (unsigned)(g_px0 * a.CA + i0 + g_part * EPU) * (unsigned)ES;      // column / channel part; the row part follows the strip per tile
    const int g_lds0 = ((g_part * EPU) >> 5) * GBLK + g_lp0 * 32 + ((g_part * EPU) & 31);
    auto g_pxy = [&](int it, int& dpx, int& dpy) {                    // pixel offset of pass `it` relative to (g_px0, g_py0): compile-time
        if (G_STEP >= TWT) { dpx = 0; dpy = it * (G_STEP / TWT); }
        else { dpx = (it * G_STEP) % TWT; dpy = (it * G_STEP) / TWT; }   // G_STEP == 16, TWT == 32: lp0 < 16, so px0 + dpx < 32 stays in the row
    };
    // X unit (per 32-channel block) u = tid + it*512: halo pixel hp = u / 8, quad part = u % 8; (hy, hx) packed in one register
    const int x_part = (int)((unsigned)tid % (unsigned)XQ);
    int x_hyx[X_IT];
#pragma unroll
    for (int it = 0; it < X_IT; ++it) {
        const int hp = (int)((unsigned)(tid + it * THREADS) / (unsigned)XQ);
        const int hy = hp / (TWT + 2);
        x_hyx[it] = (hy << 8) | (hp - hy * (TWT + 2));
    }
    auto x_alive = [&](int it) { return (it + 1) * THREADS <= X_UNITS1 || (int)((unsigned)(tid + it * THREADS) / (unsigned)XQ) < X_PIX; };
    // sources of the WCI 32-channel blocks of X (virtual concat [x0, x1]); blocks beyond C0 + C1 are zero padding
    const char* xs[WCI]; int xC[WCI], xc0[WCI];
#pragma unroll
    for (int cb = 0; cb < WCI; ++cb) {
        const int c = j0 + cb * 32;
        if (c < a.C0) { xs[cb] = static_cast<const char*>(a.x0); xC[cb] = a.C0; xc0[cb] = c; }
        else if (c < a.C0 + a.C1) { xs[cb] = static_cast<const char*>(a.x1); xC[cb] = a.C1; xc0[cb] = c - a.C0; }
        else { xs[cb] = nullptr; xC[cb] = 32; xc0[cb] = 0; }
    }
    float4 rg[G_IT], rx[WCI][X_IT];
    const size_t g_img = (size_t)a.H * a.W * a.CA * ES;
    // A tile's rows are rows of the strip: strip row v0 + r lies in image img0 = v0 / VP (r below the end of that image's pitch) or in the next one.
    // A tile that lies inside ONE image (always when seam == false -- the pitch is a multiple of TH -- and for every tile of a seamed strip that ends
    // above the next image): a scalar tile base plus per-thread constants, rows past the image end fall outside the one-image descriptor (zero
    // fill): the round-3 addressing, 3 VALU instructions per load.
    // A tile across a seam: both operands are addressed through a two-image window at image img0; a row is in the window's first image (r < lim0), in its
    // second (r >= nxt: the same per-thread constant on a second scalar base) or in the separator (no load).  Separator rows, rows past the last
    // image and columns outside the image read as zeros -- the convolution's padding for X and "no pixel" for G.
    const unsigned g_rowc = (unsigned)(g_py0 * a.W * a.CA) * (unsigned)ES;      // this thread's row part of the G offset
    auto load_tile = [&](int tile) {
        int ty, tx;
        band_tile(tile, a.tiles_x, a.tiles_y, a.band, ty, tx);
        const int v0 = ty * TH, x0 = tx * TWT;
        const int img0 = v0 / VP, vrel = v0 - img0 * VP;
        const int wrem = a.W - x0;
        if (!seam || VP - vrel > TH) {                                // no row of this tile (halo included) lies in a second image: most tiles even on a seamed strip
            const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc((void*)(static_cast<const char*>(a.g) + (size_t)img0 * g_img), 0, (int)g_img, 0x00020000);
            const int gbase = (vrel * a.W + x0) * a.CA * ES;          // rows past the image end fall outside the descriptor: zero fill
#pragma unroll
            for (int it = 0; it < G_IT; ++it) {
                int dpx, dpy;
                g_pxy(it, dpx, dpy);
                const unsigned off = g_off0 + g_rowc + (unsigned)((dpy * a.W + dpx) * a.CA) * (unsigned)ES;
                rg[it] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs_g, g_px0 + dpx < wrem ? (int)off : (int)OOB, gbase, 0));
            }
#pragma unroll
            for (int cb = 0; cb < WCI; ++cb) {
                const size_t x_img = (size_t)a.H * a.W * xC[cb] * ES;
                const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)(xs[cb] ? xs[cb] + (size_t)img0 * x_img : nullptr), 0, xs[cb] ? (int)x_img : 0, 0x00020000);
#pragma unroll
                for (int it = 0; it < X_IT; ++it) {
                    const int gy = vrel - 1 + (x_hyx[it] >> 8), gx = x0 - 1 + (x_hyx[it] & 255);
                    const bool ok = x_alive(it) && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
                    const unsigned off = ok ? (unsigned)((gy * a.W + gx) * xC[cb] + xc0[cb] + x_part * EPU) * (unsigned)ES : OOB;
                    rx[cb][it] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)off, 0, 0));
                }
            }
            return;
        }
        const int nimg = a.N - img0 < 2 ? a.N - img0 : 2;
        const int lim0 = a.H - vrel, nxt = VP - vrel;                 // tile rows [0, lim0): first image; [nxt, nxt + H): second image
        const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc((void*)(static_cast<const char*>(a.g) + (size_t)img0 * g_img), 0, (int)(g_img * nimg), 0x00020000);
        const unsigned gb0 = (unsigned)((vrel * a.W + x0) * a.CA) * (unsigned)ES;
        const unsigned gb1 = (unsigned)(((a.H - nxt) * a.W + x0) * a.CA) * (unsigned)ES;      // (may wrap: only ever added to offsets of rows >= nxt)
#pragma unroll
        for (int it = 0; it < G_IT; ++it) {
            int dpx, dpy;
            g_pxy(it, dpx, dpy);
            const int r = g_py0 + dpy;
            const bool second = r >= nxt;
            const bool ok = (r < lim0 || (second && r - nxt < a.H)) && g_px0 + dpx < wrem;
            const unsigned off = (second ? gb1 : gb0) + g_off0 + g_rowc + (unsigned)((dpy * a.W + dpx) * a.CA) * (unsigned)ES;
            rg[it] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs_g, ok ? (int)off : (int)OOB, 0, 0));
        }
#pragma unroll
        for (int cb = 0; cb < WCI; ++cb) {
            const size_t x_img = (size_t)a.H * a.W * xC[cb] * ES;
            const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)(xs[cb] ? xs[cb] + (size_t)img0 * x_img : nullptr), 0, xs[cb] ? (int)(x_img * nimg) : 0, 0x00020000);
#pragma unroll
            for (int it = 0; it < X_IT; ++it) {
                const int r = (x_hyx[it] >> 8) - 1, gx = x0 - 1 + (x_hyx[it] & 255);      // halo row relative to the tile's first row
                const bool second = r >= nxt;
                const int wrow = second ? a.H + r - nxt : vrel + r;                           // row inside the window
                const bool rok = second ? r - nxt < a.H : (unsigned)(vrel + r) < (unsigned)a.H;      // vrel + r == -1: the separator above / the strip's top edge
                const bool ok = x_alive(it) && rok && (unsigned)gx < (unsigned)a.W;
                const unsigned off = ok ? (unsigned)((wrow * a.W + gx) * xC[cb] + xc0[cb] + x_part * EPU) * (unsigned)ES : OOB;
                rx[cb][it] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)off, 0, 0));
            }
        }
    };
    auto put3 = [&](bf16_t* d, int plane_elems, const float4& raw) {       // fp32: the exact three-piece cut of conv_x3.hip::split_store; bf16: 8 elements as loaded
        if constexpr (ES == 2) { *reinterpret_cast<float4*>(d) = raw; return; }
        const unsigned x0 = __float_as_uint(raw.x), x1 = __float_as_uint(raw.y), x2 = __float_as_uint(raw.z), x3 = __float_as_uint(raw.w);
        const float r0 = raw.x - __uint_as_float(x0 & 0xFFFF0000u), r1 = raw.y - __uint_as_float(x1 & 0xFFFF0000u);
        const float r2 = raw.z - __uint_as_float(x2 & 0xFFFF0000u), r3 = raw.w - __uint_as_float(x3 & 0xFFFF0000u);
        const unsigned y0 = __float_as_uint(r0), y1 = __float_as_uint(r1), y2 = __float_as_uint(r2), y3 = __float_as_uint(r3);
        const unsigned z0 = __float_as_uint(r0 - __uint_as_float(y0 & 0xFFFF0000u)), z1 = __float_as_uint(r1 - __uint_as_float(y1 & 0xFFFF0000u));
        const unsigned z2 = __float_as_uint(r2 - __uint_as_float(y2 & 0xFFFF0000u)), z3 = __float_as_uint(r3 - __uint_as_float(y3 & 0xFFFF0000u));
        auto hp = [](unsigned lo, unsigned hi_) { return __builtin_amdgcn_perm(hi_, lo, 0x07060302u); };
        *reinterpret_cast<uint2*>(d) = make_uint2(hp(x0, x1), hp(x2, x3));
        *reinterpret_cast<uint2*>(d + plane_elems) = make_uint2(hp(y0, y1), hp(y2, y3));
        *reinterpret_cast<uint2*>(d + 2 * plane_elems) = make_uint2(hp(z0, z1), hp(z2, z3));
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int it = 0; it < G_IT; ++it) {
            put3(ldsG + g_lds0 + it * G_STEP * 32, GPL, rg[it]);
            if (do_bias) {
                if constexpr (ES == 4) { bsum4.x += rg[it].x; bsum4.y += rg[it].y; bsum4.z += rg[it].z; bsum4.w += rg[it].w; }
                else {
                    const uint4 q = __builtin_bit_cast(uint4, rg[it]);
                    const float4 lo = unpack_bf4(make_uint2(q.x, q.y)), hi4 = unpack_bf4(make_uint2(q.z, q.w));
                    bsum4.x += lo.x; bsum4.y += lo.y; bsum4.z += lo.z; bsum4.w += lo.w;
                    bsum4b.x += hi4.x; bsum4b.y += hi4.y; bsum4b.z += hi4.z; bsum4b.w += hi4.w;
                }
            }
        }
#pragma unroll
        for (int cb = 0; cb < WCI; ++cb)
#pragma unroll
            for (int it = 0; it < X_IT; ++it)
                if (x_alive(it)) put3(ldsX + (cb * X_PIX + (int)((unsigned)(tid + it * THREADS) / (unsigned)XQ)) * 32 + x_part * EPU, XPL, rx[cb][it]);
    };

    // fragment addresses (see wgrad_kernel): a 16-lane group reads a [4 pixels][16 channels] block per ds_read_b64_tr_b16
    const int gi = lane & 15, gg = lane >> 4;
    // Row-shared X fragments (round 6, STREAM == 2; 8-pixel-wide tiles owned by one pixel slice): a 16-pixel k-step is two tile ROWS, and with the rows paired as
    // (s, s + TH/2) instead of (2s, 2s + 1) the X operand of tap (ky, kx) at k-step s is rows (s + ky, s + ky + TH/2) of the halo tile -- a function of a = s + ky
    // alone.  One fragment triple F(a, kx) then serves up to three (k-step, kernel row) pairs: (TH/2 + 2) x 3 X triples per tile instead of TH/2 x 9 (TH = 8: 18
    // instead of 36; with the G triples 22 instead of 40 LDS fragment triples per 216 MFMAs).  The lane halves (hi) simply read rows TH/2 apart.
    constexpr bool ROWSHARE = STREAM == 2;
    static_assert(!ROWSHARE || (NPC == 3 && TWT == 8 && WPIX == 1 && TH % 2 == 0), "row-shared fragments: fp32, 8-pixel-wide tiles, one pixel slice");
    constexpr int HS = TH / 2;
    const int lq0 = ROWSHARE ? HS * TWT * hi + (gi >> 2) : wpix * PW + 8 * hi + (gi >> 2);
    const int pyq = lq0 / TWT, pxq = lq0 - pyq * TWT;                 // (the +4 pixels of tr8's second read and the k-step offsets below never leave the row)
    const bf16_t* gq = ldsG + wco * GBLK + lq0 * 32 + (gg & 1) * 16 + (gi & 3) * 4;
    const bf16_t* xq = ldsX + (wci * X_PIX + pyq * (TWT + 2) + pxq) * 32 + (gg & 1) * 16 + (gi & 3) * 4;
    auto tr8 = [](const bf16_t* p0) {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p0);
        const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p0 + 4 * 32));
        return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi4, 0, 1, 2, 3, 4, 5, 6, 7));
    };

    if (ps < ntiles) load_tile(ps);
    for (int tile = ps; tile < ntiles; tile += a.psplit) {
        __syncthreads();
        store_tile();
        __syncthreads();
        if (tile + a.psplit < ntiles) load_tile(tile + a.psplit);
        if constexpr (ROWSHARE) {
            // unit u = (a, kx): F(a, kx) feeds the blocks (ky, s = a - ky), 0 <= s < HS, six piece products each on acc[3 ky + kx]; an accumulator receives its
            // k-steps in ascending order.  G(s) lives in a ring of three (used at a = s, s + 1, s + 2): G(a + 1) is read in unit (a, 2) AFTER that unit's
            // ky = 2 block, the last reader of the slot's previous tenant G(a - 2).  F is double-buffered: the next unit's triple is read while this unit runs.
            constexpr int GI[6] = {0, 1, 2, 0, 1, 0};
            constexpr int XI[6] = {2, 1, 0, 1, 0, 0};
            constexpr int NU = (HS + 2) * 3;
            bf16x8 gr[3][3], xf[2][3];
            auto readG = [&](int s_, bf16x8 (&g3)[3]) {
#pragma unroll
                for (int pc = 0; pc < 3; ++pc) g3[pc] = tr8(gq + pc * GPL + s_ * TWT * 32);
            };
            auto readF = [&](int a_, int kx, bf16x8 (&x3)[3]) {
#pragma unroll
                for (int pc = 0; pc < 3; ++pc) x3[pc] = tr8(xq + pc * XPL + (a_ * (TWT + 2) + kx) * 32);
            };
            readG(0, gr[0]);
            readF(0, 0, xf[0]);
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int a_ = u / 3, kx = u % 3;
                if (u + 1 < NU) readF((u + 1) / 3, (u + 1) % 3, xf[(u + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ky = 2; ky >= 0; --ky) {
                    const int s_ = a_ - ky;
                    if (s_ >= 0 && s_ < HS) {
#pragma unroll
                        for (int q = 0; q < 6; ++q)
                            acc[3 * ky + kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gr[s_ % 3][GI[q]], xf[u & 1][XI[q]], acc[3 * ky + kx], 0, 0, 0);
                    }
                    if (ky == 2 && kx == 2 && a_ + 1 < HS) {
                        __builtin_amdgcn_sched_barrier(0);
                        readG(a_ + 1, gr[(a_ + 1) % 3]);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else
        if constexpr (NPC == 3 && STREAM == 1) {
            // Streamed fragment blocks (round 6, as conv_x3.hip x3_stage_blocks): block b = (k-step ks, tap t) is six piece products on acc[t]; the
            // fragments the NEXT block needs (its tap's three X pieces; at a k-step's last tap also the next k-step's three G pieces) are read into the
            // other half of two double buffers while the block's MFMAs run.  Per-accumulator product order as below: the same bits.
            constexpr int GI[6] = {0, 1, 2, 0, 1, 0};
            constexpr int XI[6] = {2, 1, 0, 1, 0, 0};
            constexpr int NBLK = KSB * TAPS;
            bf16x8 ga[2][3], xb[2][3];
            auto readG = [&](int ks, bf16x8 (&g3)[3]) {
#pragma unroll
                for (int pc = 0; pc < 3; ++pc) g3[pc] = tr8(gq + pc * GPL + ks * 16 * 32);
            };
            auto readX = [&](int ks, int t, bf16x8 (&x3)[3]) {
                const int lrel = ks * 16, dy0 = lrel / TWT, dxp = lrel - dy0 * TWT;
                const int xoff = (dy0 + t / 3) * (TWT + 2) + dxp + t % 3;
#pragma unroll
                for (int pc = 0; pc < 3; ++pc) x3[pc] = tr8(xq + pc * XPL + xoff * 32);
            };
            readG(0, ga[0]);
            readX(0, 0, xb[0]);
#pragma unroll
            for (int b = 0; b < NBLK; ++b) {
                const int ks = b / TAPS, t = b % TAPS;
                if (b + 1 < NBLK) {
                    const int ks1 = (b + 1) / TAPS, t1 = (b + 1) % TAPS;
                    readX(ks1, t1, xb[(b + 1) & 1]);
                    if (ks1 != ks) readG(ks1, ga[ks1 & 1]);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 6; ++q) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga[ks & 1][GI[q]], xb[b & 1][XI[q]], acc[t], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else
#pragma unroll
        for (int ks = 0; ks < KSB; ++ks) {
            const int lrel = ks * 16;
            const int dy0 = lrel / TWT, dxp = lrel - dy0 * TWT;
            bf16x8 ga[NPC];
#pragma unroll
            for (int pc = 0; pc < NPC; ++pc) ga[pc] = tr8(gq + pc * GPL + lrel * 32);
#pragma unroll
            for (int t0 = 0; t0 < TAPS; t0 += 3) {
                bf16x8 xb[3][NPC];
#pragma unroll
                for (int tt = 0; tt < 3; ++tt) {
                    const int xoff = (dy0 + t0 / 3) * (TWT + 2) + dxp + tt;
#pragma unroll
                    for (int pc = 0; pc < NPC; ++pc) xb[tt][pc] = tr8(xq + pc * XPL + xoff * 32);
                }
                if constexpr (NPC == 3) {
                    constexpr int GI[6] = {0, 1, 2, 0, 1, 0};
                    constexpr int XI[6] = {2, 1, 0, 1, 0, 0};
#pragma unroll
                    for (int q = 0; q < 6; ++q)
#pragma unroll
                        for (int tt = 0; tt < 3; ++tt)
                            acc[t0 + tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga[GI[q]], xb[tt][XI[q]], acc[t0 + tt], 0, 0, 0);
                } else {
#pragma unroll
                    for (int tt = 0; tt < 3; ++tt) acc[t0 + tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga[0], xb[tt][0], acc[t0 + tt], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);                    // keep the next kernel row's reads below: 256-VGPR budget
            }
        }
    }

    // ---- reduce the WPIX pixel slices through LDS, one tap at a time; slice 0 writes the partial ----------------------------------
    __syncthreads();
    float* red = lds;                                                  // [(WPIX-1)][WCO*WCI][16][64]
    const size_t pbase = (size_t)ps * TAPS * a.CA * a.CBp;
    const int wt = wci * WCO + wco;
#pragma unroll
    for (int t = 0; t < TAPS; ++t) {
        if (WPIX > 1) {
            if (wpix > 0) {
#pragma unroll
                for (int i = 0; i < 16; ++i) red[(((wpix - 1) * (WCO * WCI) + wt) * 16 + i) * 64 + lane] = acc[t][i];
            }
            __syncthreads();
            if (wpix == 0) {
                for (int w = 0; w < WPIX - 1; ++w)
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[t][i] += red[((w * (WCO * WCI) + wt) * 16 + i) * 64 + lane];
            }
            __syncthreads();
        }
        if (wpix == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = (i & 3) + 8 * (i >> 2) + 4 * hi;
                a.part[pbase + ((size_t)t * a.CA + i0 + wco * 32 + row) * a.CBp + j0 + wci * 32 + m] = acc[t][i];
            }
        }
    }
    if (do_bias) {                                                     // thread -> channel quad g_part; 512 / G_Q threads per quad, fixed order
        __syncthreads();
        if constexpr (ES == 4) reinterpret_cast<float4*>(red)[tid] = bsum4;
        else { reinterpret_cast<float4*>(red)[2 * tid] = bsum4; reinterpret_cast<float4*>(red)[2 * tid + 1] = bsum4b; }
        __syncthreads();
        if (tid < COB) {
            const int q = tid / EPU, comp = tid % EPU;
            float s_ = 0.f;
            for (int k = 0; k < THREADS / G_Q; ++k) s_ += red[(q + k * G_Q) * EPU + comp];
            a.bpart[(size_t)ps * a.CA + i0 + tid] = s_;
        }
    }
}

// =============================================================================================================================
// wgrad8d_kernel (round 4) -- wgrad8_kernel for bf16 tensors with BOTH operand tiles staged by LDS-DMA.
// bf16 operands need no conversion on the way in, yet wgrad8_kernel<bf16_t> moved every tile HBM -> VGPR -> ds_write: 109 us of staging stores
// and 65 us of exposed loads beside 272 us of MFMA per launch of the 128 x 64 blocks (profiles/r03_ab_notes.md), two barriers per tile.  Here a
// tile's 16-byte units travel straight to LDS (buffer_load_dwordx4 ... lds, as conv_bfd.hip): one wave instruction = 1 KiB = 16 pixels x 64 B
// of one 32-channel block, landing linearly -- which IS the [32-channel block][pixel][32] plane layout the ds_read_b64_tr_b16 fragment reads
// want.  Two tile buffers: the DMA of tile t+1 is issued behind the barrier that publishes tile t and flies under tile t's MFMAs; one barrier
// per tile, no staging registers, no VALU on the data.  Zero padding (image border, separator rows of the strip, channel padding of X) =
// out-of-range buffer offsets.  The bias gradient (column sums of G) is read back from the landed G tile by the jb == 0 blocks.
// Tiles: TH x TWT pixels over the virtual-row strip, as wgrad8_kernel; a block of X is padded to whole 1 KiB pieces so that a DMA instruction
// never straddles two source tensors (virtual concat [x0, x1]).
__device__ __forceinline__ void wg_dma16(i32x4_t rsrc, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_dst) : "memory");
}

template <int WCO, int WCI, int WPIX, int TH, int TWT>
__global__ __launch_bounds__(512, 2) void wgrad8d_kernel(const WgradArgs a) {
    static_assert(WCO * WCI * WPIX == 8, "8 waves");
    static_assert(TWT == 8 || TWT == 16 || TWT == 32, "tile width");
    constexpr int WAVES = 8, THREADS = 512, TAPS = 9;
    constexpr int COB = 32 * WCO, JBK = 32 * WCI, TPIX = TH * TWT, PW = TPIX / WPIX, KSB = PW / 16;
    static_assert(PW % 16 == 0 && PW >= 16, "a wave's pixel slice is a whole number of 16-pixel k-steps");
    constexpr int HW = TWT + 2, X_PIX = (TH + 2) * HW;
    static_assert((TPIX * 4) % 64 == 0, "a block of G is whole DMA pieces");
    constexpr int GB_PIECES = TPIX * 4 / 64, XB_PIECES = (X_PIX * 4 + 63) / 64;      // 1 KiB pieces of one 32-channel block (4 units of 16 B per pixel)
    constexpr int G_PIECES = WCO * GB_PIECES, X_PIECES = WCI * XB_PIECES;
    constexpr int G_IT = (G_PIECES + WAVES - 1) / WAVES, X_IT = (X_PIECES + WAVES - 1) / WAVES;
    constexpr int GBLK = GB_PIECES * 512, XBLK = XB_PIECES * 512;                   // bf16 elements of one block plane
    constexpr int BUF_BYTES = (G_PIECES + X_PIECES) * 1024;
    extern __shared__ __attribute__((aligned(16))) float lds[];                    // [2 buffers][G: WCO blocks | X: WCI blocks]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 31, hi = lane >> 5;
    const int wco = wave % WCO, wci = (wave / WCO) % WCI, wpix = wave / (WCO * WCI);
    const int IB = a.CA / COB, JBn = a.CBp / JBK;
    int bid = xcd_block(a.xcd);
    const int jb = bid % JBn; bid /= JBn;
    const int ib = bid % IB;
    const int ps = bid / IB;
    const int i0 = ib * COB, j0 = jb * JBK;
    const bool do_bias = a.bpart != nullptr && jb == 0;

    f32x16 acc[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    float4 bsum4 = make_float4(0.f, 0.f, 0.f, 0.f), bsum4b = make_float4(0.f, 0.f, 0.f, 0.f);

    const int ntiles = a.tiles_x * a.tiles_y;
    const int VP = a.vp;
    const bool seam = VP % TH != 0;
    constexpr unsigned OOB = 0xFFFFFFF0u;
    const unsigned lds_base = (unsigned)__builtin_amdgcn_readfirstlane((int)(size_t)(__attribute__((address_space(3))) float*)lds);

    // sources of the WCI 32-channel blocks of X (virtual concat [x0, x1]); blocks beyond C0 + C1 are zero padding (no descriptor range at all)
    const char* xs[WCI]; int xC[WCI], xc0[WCI];
#pragma unroll
    for (int cb = 0; cb < WCI; ++cb) {
        const int c = j0 + cb * 32;
        if (c < a.C0) { xs[cb] = static_cast<const char*>(a.x0); xC[cb] = a.C0; xc0[cb] = c; }
        else if (c < a.C0 + a.C1) { xs[cb] = static_cast<const char*>(a.x1); xC[cb] = a.C1; xc0[cb] = c - a.C0; }
        else { xs[cb] = nullptr; xC[cb] = 32; xc0[cb] = 0; }
    }
    // ---- which unit this lane moves in each of its wave's pieces: pieces are dealt round-robin (a duplicate rewrites identical bytes) --------
    // G piece p: block p / GB_PIECES, units 64 (p % GB_PIECES) ..: unit -> pixel (unit >> 2) of the tile, 8-channel quarter unit & 3
    unsigned g_c[G_IT]; int g_rc[G_IT];
#pragma unroll
    for (int it = 0; it < G_IT; ++it) {
        const int piece = (wave + it * WAVES) % G_PIECES;
        const int blk = piece / GB_PIECES, ub = (piece - blk * GB_PIECES) * 64 + lane;
        const int pix = ub >> 2, q = ub & 3;
        const int py = pix / TWT, px = pix - py * TWT;
        g_c[it] = (unsigned)((py * a.W + px) * a.CA + i0 + blk * 32 + q * 8) * 2u;
        g_rc[it] = (py << 8) | px;
    }
    // X piece p: block p / XB_PIECES; unit -> halo pixel (unit >> 2) = (hy, hx), quarter unit & 3; units past the halo tile move nothing
    unsigned x_c[X_IT]; int x_rc[X_IT];
#pragma unroll
    for (int it = 0; it < X_IT; ++it) {
        const int piece = (wave + it * WAVES) % X_PIECES;
        const int blk = piece / XB_PIECES, ub = (piece - blk * XB_PIECES) * 64 + lane;
        const int hp = ub >> 2, q = ub & 3;
        const int hy = hp / HW, hx = hp - hy * HW;
        int C = xC[0], c0 = xc0[0];
#pragma unroll
        for (int cb = 1; cb < WCI; ++cb) if (blk == cb) { C = xC[cb]; c0 = xc0[cb]; }
        x_c[it] = (unsigned)(((hy - 1) * a.W + (hx - 1)) * C + c0 + q * 8) * 2u;      // (may wrap: always added to a base that makes it a valid offset)
        x_rc[it] = hp < X_PIX ? ((hy << 8) | hx) : -1;
    }
    const size_t g_img = (size_t)a.H * a.W * a.CA * 2;

    // DMA of one tile into buffer `buf` (see wgrad8_kernel::load_tile for the strip / seam rules)
    auto issue = [&](int tile, int buf) {
        int ty, tx;
        band_tile(tile, a.tiles_x, a.tiles_y, a.band, ty, tx);
        const int v0 = ty * TH, x0 = tx * TWT;
        const int img0 = v0 / VP, vrel = v0 - img0 * VP;
        const int wrem = a.W - x0;
        const bool single = !seam || VP - vrel > TH;                  // no row of the tile (halo included) lies in a second image
        const int nimg = single ? 1 : (a.N - img0 < 2 ? a.N - img0 : 2);
        const int lim0 = a.H - vrel, nxt = single ? (1 << 20) : VP - vrel;      // tile rows [0, lim0): first image; [nxt, nxt + H): second image
        const int row1 = single ? 0 : a.H - nxt;                      // window row of tile row 0 if it were in the second image (may be negative)
        const unsigned dst0 = lds_base + (unsigned)(buf * BUF_BYTES);
        {
            const unsigned long long gb = (unsigned long long)(static_cast<const char*>(a.g) + (size_t)img0 * g_img);
            const i32x4_t rs = {(int)(unsigned)gb, (int)((unsigned)(gb >> 32) & 0xFFFFu), (int)(unsigned)(g_img * (size_t)nimg), 0x00020000};
            const unsigned b0 = (unsigned)((vrel * a.W + x0) * a.CA) * 2u, b1 = (unsigned)((row1 * a.W + x0) * a.CA) * 2u;
#pragma unroll
            for (int it = 0; it < G_IT; ++it) {
                const int piece = (wave + it * WAVES) % G_PIECES;
                const int r = g_rc[it] >> 8, px = g_rc[it] & 255;
                const bool second = r >= nxt;
                const bool ok = (r < lim0 || (second && r - nxt < a.H)) && px < wrem;
                wg_dma16(rs, ok ? (second ? b1 : b0) + g_c[it] : OOB, dst0 + (unsigned)(piece * 1024));
            }
        }
#pragma unroll
        for (int it = 0; it < X_IT; ++it) {
            const int piece = (wave + it * WAVES) % X_PIECES;
            const int blk = piece / XB_PIECES;                        // wave-uniform
            const char* src = xs[0]; int C = xC[0];
#pragma unroll
            for (int cb = 1; cb < WCI; ++cb) if (blk == cb) { src = xs[cb]; C = xC[cb]; }
            const size_t x_img = (size_t)a.H * a.W * C * 2;
            const unsigned long long xb = (unsigned long long)(src ? src + (size_t)img0 * x_img : nullptr);
            const i32x4_t rs = {(int)(unsigned)xb, (int)((unsigned)(xb >> 32) & 0xFFFFu), src ? (int)(unsigned)(x_img * (size_t)nimg) : 0, 0x00020000};
            const unsigned b0 = (unsigned)((vrel * a.W + x0) * C) * 2u, b1 = (unsigned)((row1 * a.W + x0) * C) * 2u;
            const int r = (x_rc[it] >> 8) - 1, gx = x0 - 1 + (x_rc[it] & 255);
            const bool second = r >= nxt;
            const bool rok = second ? r - nxt < a.H : (unsigned)(vrel + r) < (unsigned)a.H;
            const bool ok = x_rc[it] >= 0 && rok && (unsigned)gx < (unsigned)a.W;
            wg_dma16(rs, ok ? (second ? b1 : b0) + x_c[it] : OOB, dst0 + (unsigned)((G_PIECES + piece) * 1024));
        }
    };

    // fragment addresses (see wgrad_kernel): a 16-lane group reads a [4 pixels][16 channels] block per ds_read_b64_tr_b16
    const int gi = lane & 15, gg = lane >> 4;
    const int lq0 = wpix * PW + 8 * hi + (gi >> 2);
    const int pyq = lq0 / TWT, pxq = lq0 - pyq * TWT;
    const bf16_t* ldsG = reinterpret_cast<const bf16_t*>(lds);
    const bf16_t* gq0 = ldsG + wco * GBLK + lq0 * 32 + (gg & 1) * 16 + (gi & 3) * 4;
    const bf16_t* xq0 = ldsG + G_PIECES * 512 + wci * XBLK + (pyq * HW + pxq) * 32 + (gg & 1) * 16 + (gi & 3) * 4;
    auto tr8 = [](const bf16_t* p0) {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p0);
        const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p0 + 4 * 32));
        return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi4, 0, 1, 2, 3, 4, 5, 6, 7));
    };
    // bias sums: thread -> 8-channel group g_part of the block's COB channels (as wgrad8_kernel), pixels tid / G_Q + k * (512 / G_Q)
    constexpr int G_Q = COB / 8, B_STEP = THREADS / G_Q, B_IT = TPIX / B_STEP;
    static_assert(TPIX % B_STEP == 0, "bias pass covers the tile");
    const int g_part = tid % G_Q;
    const bf16_t* bq0 = ldsG + (g_part >> 2) * GBLK + (tid / G_Q) * 32 + (g_part & 3) * 8;

    int buf = 0;
    if (ps < ntiles) issue(ps, 0);
    for (int tile = ps; tile < ntiles; tile += a.psplit) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // this wave's pieces of the tile have landed ...
        __syncthreads();                                              // ... and everybody else's; the other buffer is no longer read
        // (waves 4-7 issuing their pieces behind their first k-step, as conv_bfd_kernel does, measured +1.3 % here: profiles/r04_ab_notes.md)
        if (tile + a.psplit < ntiles) issue(tile + a.psplit, buf ^ 1);
        const int bo = buf * (BUF_BYTES / 2);                        // buffer offset in bf16 elements
        if (do_bias) {
#pragma unroll
            for (int k = 0; k < B_IT; ++k) {
                const uint4 q = *reinterpret_cast<const uint4*>(bq0 + bo + k * B_STEP * 32);
                const float4 lo = unpack_bf4(make_uint2(q.x, q.y)), hi4 = unpack_bf4(make_uint2(q.z, q.w));
                bsum4.x += lo.x; bsum4.y += lo.y; bsum4.z += lo.z; bsum4.w += lo.w;
                bsum4b.x += hi4.x; bsum4b.y += hi4.y; bsum4b.z += hi4.z; bsum4b.w += hi4.w;
            }
        }
        const bf16_t* gq = gq0 + bo;
        const bf16_t* xq = xq0 + bo;
#pragma unroll
        for (int ks = 0; ks < KSB; ++ks) {
            const int lrel = ks * 16;
            const int dy0 = lrel / TWT, dxp = lrel - dy0 * TWT;
            const bf16x8 ga = tr8(gq + lrel * 32);
#pragma unroll
            for (int t0 = 0; t0 < TAPS; t0 += 3) {
                bf16x8 xb[3];
#pragma unroll
                for (int tt = 0; tt < 3; ++tt) xb[tt] = tr8(xq + ((dy0 + t0 / 3) * HW + dxp + tt) * 32);
#pragma unroll
                for (int tt = 0; tt < 3; ++tt) acc[t0 + tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga, xb[tt], acc[t0 + tt], 0, 0, 0);
            }
        }
        buf ^= 1;
    }

    // ---- reduce the WPIX pixel slices through LDS, one tap at a time; slice 0 writes the partial (as wgrad8_kernel) -----------------------
    __syncthreads();
    float* red = lds;                                                  // [(WPIX-1)][WCO*WCI][16][64]
    const size_t pbase = (size_t)ps * TAPS * a.CA * a.CBp;
    const int wt = wci * WCO + wco;
#pragma unroll
    for (int t = 0; t < TAPS; ++t) {
        if (WPIX > 1) {
            if (wpix > 0) {
#pragma unroll
                for (int i = 0; i < 16; ++i) red[(((wpix - 1) * (WCO * WCI) + wt) * 16 + i) * 64 + lane] = acc[t][i];
            }
            __syncthreads();
            if (wpix == 0) {
                for (int w = 0; w < WPIX - 1; ++w)
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[t][i] += red[((w * (WCO * WCI) + wt) * 16 + i) * 64 + lane];
            }
            __syncthreads();
        }
        if (wpix == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = (i & 3) + 8 * (i >> 2) + 4 * hi;
                a.part[pbase + ((size_t)t * a.CA + i0 + wco * 32 + row) * a.CBp + j0 + wci * 32 + m] = acc[t][i];
            }
        }
    }
    if (do_bias) {                                                     // thread -> 8-channel group g_part; 512 / G_Q threads per group, fixed order
        __syncthreads();
        reinterpret_cast<float4*>(red)[2 * tid] = bsum4; reinterpret_cast<float4*>(red)[2 * tid + 1] = bsum4b;
        __syncthreads();
        if (tid < COB) {
            const int q = tid / 8, comp = tid % 8;
            float s_ = 0.f;
            for (int k = 0; k < THREADS / G_Q; ++k) s_ += red[(q + k * G_Q) * 8 + comp];
            a.bpart[(size_t)ps * a.CA + i0 + tid] = s_;
        }
    }
}

// block shape of wgrad8_kernel for a layer (CA out-channels of G, CBp padded in-channels of X); false: the layer stays on wgrad_kernel.
// bf16 inputs keep one plane per operand (a third of the LDS of the three-piece tiles), so every block shape takes 256-pixel tiles:
// halo overhead of X 1.33x instead of 1.6x and a quarter of the barriers per pixel.  The 128 x 64 blocks take tiles 8 pixels wide (round 4):
// the levels they run on are 532, 266, 133 pixels wide -- 536 / 272 / 136 columns of 8-wide tiles against 544 / 288 / 160 of 32-wide ones --
// and the halo of a 64-pixel tile shrinks from 4 x 34 to 10 x 10 pixels.
// The 128 x 64 blocks of bf16 tensors run wgrad8d_kernel (both tiles by LDS-DMA, two tile buffers; ELD_WGRAD_DMA=0: back on wgrad8_kernel<bf16_t>
// with its 256-pixel register-staged tiles).  The smaller blocks stay on wgrad8_kernel: they are the 32 / 64-channel layers, at the HBM roofline
// already (conv1_2: 3.5 GB per launch in 0.65 ms = 5.4 TB/s); their DMA variants (16 x 16 tiles) measured 3 ... 17 % slower.
static int wgrad8_dma() {
    static const int on = [] { const char* e = getenv("ELD_WGRAD_DMA"); return e ? (int)(atoi(e) != 0) : 1; }();
    return on && !(debug_kernel_mask(-1) & 16);      // test hook (eld_debug_kernel_mask bit 4): back on the register-staged kernel
}
// =============================================================================================================================
// wgradt8_kernel (round 6) -- the transposed convolutions' weight gradient, dW[ci][tap][co] = sum_p in[p][ci] * dout[2p + tap][co] (fp32 inputs, three-piece
// scheme), re-blocked like wgrad8_kernel: ONE 8-wave workgroup per CU, output block 128 (ci) x 64 (co) x 4 taps, a wave owns a 32 x 32 tile of all four taps
// (64 accumulator registers) and walks the 8 x 8-pixel tiles of its pixel slice.  Against wgrad_kernel<float, CONV_GATHER2X2, 2, 2, ALG_X3> (64 x 32 blocks,
// 4 waves): a staged value feeds twice the MFMAs (12 operand cuts per thread per 96 MFMAs instead of per 48), one workgroup per CU instead of two.
// dout has no halo here: every dout pixel belongs to exactly one (pixel, tap), so the X tile is four tap planes of the tile's 64 pixels, [piece][32-ch block][tap]
// [pixel][32] -- each plane reads exactly like the G tile.  Tiles never straddle images.  Partials / bias column sums / reduction as wgrad_kernel's.
// C0 % 64 == 0 (no channel padding), CA % 128 == 0.
template <int DUMMY>
__global__ __launch_bounds__(512, 2) void wgradt8_kernel(const WgradArgs a) {
    constexpr int WCO = 4, WCI = 2, TH = 8, TWT = 8, TPIX = TH * TWT, THREADS = 512, TAPS = 4, NPC = 3;
    constexpr int COB = 32 * WCO, JBK = 32 * WCI, KSB = TPIX / 16;
    constexpr int GBLK = TPIX * 32 + 32;                              // one pad row per 32-channel block of G (see wgrad8_kernel)
    constexpr int GPL = GBLK * WCO, X_PIX = TAPS * TPIX, XPL = X_PIX * JBK;
    constexpr int G_IT = TPIX * (COB / 4) / THREADS, X_IT = X_PIX * 8 / THREADS;      // 4 and 4 (per 32-channel block of X)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    bf16_t* ldsG = reinterpret_cast<bf16_t*>(lds);                    // [3][WCO][GBLK]
    bf16_t* ldsX = ldsG + NPC * GPL;                                  // [3][WCI][X_PIX][32]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = lane & 31, hi = lane >> 5;
    const int wco = wave % WCO, wci = wave / WCO;
    const int IB = a.CA / COB, JBn = a.CBp / JBK;
    int bid = xcd_block(a.xcd);
    const int jb = bid % JBn; bid /= JBn;
    const int ib = bid % IB;
    const int ps = bid / IB;
    const int i0 = ib * COB, j0 = jb * JBK;
    const bool xsum_on = a.xbpart != nullptr && ib == 0;
    const int Wx = 2 * a.W, CB = a.C0;

    f32x16 acc[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    float4 xs4[WCI];
#pragma unroll
    for (int cb = 0; cb < WCI; ++cb) xs4[cb] = make_float4(0.f, 0.f, 0.f, 0.f);

    const int tiles_per_img = a.tiles_x * a.tiles_y, ntiles = tiles_per_img * a.N;
    constexpr unsigned OOB = 0xFFFFFFF0u;
    // staging constants: G unit u = tid + it * 512 -> pixel lp = u / 32 (= tid / 32 + 16 it), channel quad u % 32 (the same for every it);
    // X unit (per 32-channel block) u -> (tap, pixel) = u / 8 (= tid / 8 + 64 it: tap = it), quad u % 8
    const int g_part = tid & 31, g_lp0 = tid >> 5, x_part = tid & 7, x_lp = tid >> 3;
    const int x_py = x_lp / TWT, x_px = x_lp - x_py * TWT;
    float4 rg[G_IT], rx[WCI][X_IT];
    const size_t g_img = (size_t)a.H * a.W * a.CA * 4, x_img = (size_t)4 * a.H * a.W * CB * 4;
    auto load_tile = [&](int tile) {
        const int img = tile / tiles_per_img;
        int ty, tx;
        band_tile(tile - img * tiles_per_img, a.tiles_x, a.tiles_y, a.band, ty, tx);
        const int y0 = ty * TH, x0 = tx * TWT;
        const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc((void*)(static_cast<const char*>(a.g) + (size_t)img * g_img), 0, (int)g_img, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)(static_cast<const char*>(a.x0) + (size_t)img * x_img), 0, (int)x_img, 0x00020000);
#pragma unroll
        for (int it = 0; it < G_IT; ++it) {
            const int lp = g_lp0 + 16 * it, py = lp / TWT, px = lp - py * TWT;
            const int gy = y0 + py, gx = x0 + px;
            const unsigned off = (gy < a.H && gx < a.W) ? (unsigned)(((gy * a.W + gx) * a.CA + i0 + g_part * 4) * 4) : OOB;
            rg[it] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs_g, (int)off, 0, 0));
        }
        const bool pok = y0 + x_py < a.H && x0 + x_px < a.W;
#pragma unroll
        for (int cb = 0; cb < WCI; ++cb)
#pragma unroll
            for (int it = 0; it < X_IT; ++it) {                       // it = tap
                const int gy = 2 * (y0 + x_py) + (it >> 1), gx = 2 * (x0 + x_px) + (it & 1);
                const unsigned off = pok ? (unsigned)(((gy * Wx + gx) * CB + j0 + cb * 32 + x_part * 4) * 4) : OOB;
                rx[cb][it] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)off, 0, 0));
            }
    };
    auto put3 = [&](bf16_t* d, int plane_elems, const float4& raw) {       // the exact three-piece cut of conv_x3_dev.h split_store
        const unsigned x0 = __float_as_uint(raw.x), x1 = __float_as_uint(raw.y), x2 = __float_as_uint(raw.z), x3 = __float_as_uint(raw.w);
        const float r0 = raw.x - __uint_as_float(x0 & 0xFFFF0000u), r1 = raw.y - __uint_as_float(x1 & 0xFFFF0000u);
        const float r2 = raw.z - __uint_as_float(x2 & 0xFFFF0000u), r3 = raw.w - __uint_as_float(x3 & 0xFFFF0000u);
        const unsigned y0 = __float_as_uint(r0), y1 = __float_as_uint(r1), y2 = __float_as_uint(r2), y3 = __float_as_uint(r3);
        const unsigned z0 = __float_as_uint(r0 - __uint_as_float(y0 & 0xFFFF0000u)), z1 = __float_as_uint(r1 - __uint_as_float(y1 & 0xFFFF0000u));
        const unsigned z2 = __float_as_uint(r2 - __uint_as_float(y2 & 0xFFFF0000u)), z3 = __float_as_uint(r3 - __uint_as_float(y3 & 0xFFFF0000u));
        auto hp = [](unsigned lo, unsigned hi_) { return __builtin_amdgcn_perm(hi_, lo, 0x07060302u); };
        *reinterpret_cast<uint2*>(d) = make_uint2(hp(x0, x1), hp(x2, x3));
        *reinterpret_cast<uint2*>(d + plane_elems) = make_uint2(hp(y0, y1), hp(y2, y3));
        *reinterpret_cast<uint2*>(d + 2 * plane_elems) = make_uint2(hp(z0, z1), hp(z2, z3));
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int it = 0; it < G_IT; ++it)
            put3(ldsG + ((g_part * 4) >> 5) * GBLK + (g_lp0 + 16 * it) * 32 + ((g_part * 4) & 31), GPL, rg[it]);
#pragma unroll
        for (int cb = 0; cb < WCI; ++cb)
#pragma unroll
            for (int it = 0; it < X_IT; ++it) {
                put3(ldsX + (cb * X_PIX + it * TPIX + x_lp) * 32 + x_part * 4, XPL, rx[cb][it]);
                if (xsum_on) { xs4[cb].x += rx[cb][it].x; xs4[cb].y += rx[cb][it].y; xs4[cb].z += rx[cb][it].z; xs4[cb].w += rx[cb][it].w; }      // (out-of-image units loaded zeros)
            }
    };
    // fragment addresses (wgrad8_kernel): a 16-lane group reads a [4 pixels][16 channels] block per ds_read_b64_tr_b16
    const int gi = lane & 15, gg = lane >> 4;
    const int lq0 = 8 * hi + (gi >> 2);
    const bf16_t* gq = ldsG + wco * GBLK + lq0 * 32 + (gg & 1) * 16 + (gi & 3) * 4;
    const bf16_t* xq = ldsX + (wci * X_PIX + lq0) * 32 + (gg & 1) * 16 + (gi & 3) * 4;
    auto tr8 = [](const bf16_t* p0) {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p0);
        const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p0 + 4 * 32));
        return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi4, 0, 1, 2, 3, 4, 5, 6, 7));
    };

    if (ps < ntiles) load_tile(ps);
    for (int tile = ps; tile < ntiles; tile += a.psplit) {
        __syncthreads();
        store_tile();
        __syncthreads();
        if (tile + a.psplit < ntiles) load_tile(tile + a.psplit);
#pragma unroll
        for (int ks = 0; ks < KSB; ++ks) {
            bf16x8 ga[NPC];
#pragma unroll
            for (int pc = 0; pc < NPC; ++pc) ga[pc] = tr8(gq + pc * GPL + ks * 16 * 32);
#pragma unroll
            for (int t0 = 0; t0 < TAPS; t0 += 2) {
                bf16x8 xb[2][NPC];
#pragma unroll
                for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                    for (int pc = 0; pc < NPC; ++pc) xb[tt][pc] = tr8(xq + pc * XPL + ((t0 + tt) * TPIX + ks * 16) * 32);
                constexpr int GI[6] = {0, 1, 2, 0, 1, 0};
                constexpr int XI[6] = {2, 1, 0, 1, 0, 0};
#pragma unroll
                for (int q = 0; q < 6; ++q)
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt)
                        acc[t0 + tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga[GI[q]], xb[tt][XI[q]], acc[t0 + tt], 0, 0, 0);
            }
        }
    }
    // ---- the partial of this block -------------------------------------------------------------------------------------
    const size_t pbase = (size_t)ps * TAPS * a.CA * a.CBp;
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = (i & 3) + 8 * (i >> 2) + 4 * hi;
            a.part[pbase + ((size_t)t * a.CA + i0 + wco * 32 + row) * a.CBp + j0 + wci * 32 + m] = acc[t][i];
        }
    if (xsum_on) {                                                    // block-uniform: column sums of dout over this block's pixels; 64 threads hold each channel quad, fixed-order combine
        __syncthreads();
        float4* r = reinterpret_cast<float4*>(lds);                   // [WCI][512]
#pragma unroll
        for (int cb = 0; cb < WCI; ++cb) r[cb * THREADS + tid] = xs4[cb];
        __syncthreads();
        if (tid < JBK) {
            const int cb = tid >> 5, c = tid & 31, quad = c >> 2, comp = c & 3;
            float s_ = 0.f;
            for (int k = 0; k < THREADS / 8; ++k) s_ += reinterpret_cast<const float*>(&r[cb * THREADS + quad + 8 * k])[comp];
            a.xbpart[(size_t)ps * a.CBp + j0 + tid] = s_;
        }
    }
}

bool wgradt8_takes(int CA, int CB) {
    static const int on = [] { const char* e = getenv("ELD_WGRADT8"); return e ? atoi(e) : 1; }();      // ELD_WGRADT8=0: the transposed convs' weight gradient stays on wgrad_kernel
    return on && CA % 128 == 0 && CB % 64 == 0;
}
int wgradt8_ntiles(int N, int H, int W) { return ((W + 7) / 8) * ((H + 7) / 8) * N; }

static int launch_wt8(WgradArgs a, hipStream_t st) {
    a.xcd = eld_xcd_mask() & XCD_WGRAD8;
    a.band = eld_tile_band();
    a.tiles_x = (a.W + 7) / 8;
    a.tiles_y = (a.H + 7) / 8;
    const size_t lds_bytes = (size_t)(3 * 4 * (64 * 32 + 32) + 3 * 2 * 256 * 32) * 2;
    const long long blocks = (long long)(a.CA / 128) * (a.CBp / 64) * a.psplit;
    if (blocks <= 0) return 0;
    if ((size_t)4 * a.H * a.W * a.C0 * 4 >= 0xFFFFFFF0ull || (size_t)a.H * a.W * a.CA * 4 >= 0xFFFFFFF0ull) return ELD_ENOTSUP;
    auto kern = wgradt8_kernel<0>;
    static EldAttrOnce once;
    { const int rc = once.ensure(kern, lds_bytes); if (rc) return rc; }
    ELD_LAUNCH(kern, dim3((unsigned)blocks), dim3(512), lds_bytes, st, a);
    ELD_LAUNCH_CHECK();
    return 0;
}

bool wgrad8_shape(int CA, int CBp, int& COB, int& JBK, int& TH, int& TWo, bool bf16) {
    if (CA % 32 || CBp % 32) return false;
    TWo = 32;
    if (CA % 128 == 0 && CBp % 64 == 0) { COB = 128; JBK = 64; TH = 2; }
    else if (CA % 64 == 0 && CBp % 64 == 0) { COB = 64; JBK = 64; TH = 2; }      // fp32: TH = 4 spills 9 registers: 3.06 vs 2.79 ms (measured)
    else if (CA % 64 == 0) { COB = 64; JBK = 32; TH = 4; }
    else if (CBp % 64 == 0) { COB = 32; JBK = 64; TH = 4; }
    else { COB = 32; JBK = 32; TH = 4; }
    if (bf16) TH = 8;
    if (bf16 && wgrad8_dma() && COB == 128) { TH = 16; TWo = 8; return true; }
    // the 128 x 64 blocks are the >= 128-channel layers = the ragged levels (356 x 532 and below): 8-wide tiles of the same pixel count.  The
    // 32 / 64-channel layers sit on 2128- and 1064-pixel rows (nothing to win) and are bound by their staging traffic: 8-wide tiles measured
    // 3 ... 19 % slower there (profiles/r04_ab_notes.md)
    if (COB == 128) { TH = TH * 4; TWo = 8; }
    return true;
}

int wgrad8_ntiles(int CA, int CBp, int N, int H, int W, bool bf16) {
    int COB, JBK, TH, TWo;
    if (!wgrad8_shape(CA, CBp, COB, JBK, TH, TWo, bf16)) return 0;
    const int VP = vrow_pitch(N, H, TH);
    return ((W + TWo - 1) / TWo) * ((vrow_extent(N, H, VP) + TH - 1) / TH);
}

template <typename T, int WCO, int WCI, int WPIX, int TH, int TWT, int STREAM = 0>
static int launch_w8(WgradArgs a, hipStream_t st) {
    a.xcd = eld_xcd_mask() & XCD_WGRAD8;
    a.band = eld_tile_band();
    constexpr int COB = 32 * WCO, JBK = 32 * WCI;
    a.vp = vrow_pitch(a.N, a.H, TH);
    a.tiles_x = (a.W + TWT - 1) / TWT;
    a.tiles_y = (vrow_extent(a.N, a.H, a.vp) + TH - 1) / TH;
    size_t lds_bytes = (size_t)((TH * TWT * 32 + (sizeof(T) == 4 ? 32 : 0)) * (COB / 32) + (TH + 2) * (TWT + 2) * JBK) * (sizeof(T) == 4 ? 6 : 2);
    const size_t red_bytes = (size_t)7 * 16 * 64 * sizeof(float) * 1;       // WPIX-1 <= 7 slices of WCO*WCI*WPIX/... tiles: (WPIX-1)*WCO*WCI <= 7
    if (lds_bytes < red_bytes) lds_bytes = red_bytes;
    const long long blocks = (long long)(a.CA / COB) * (a.CBp / JBK) * a.psplit;
    if (blocks <= 0) return 0;
    auto kern = wgrad8_kernel<T, WCO, WCI, WPIX, TH, TWT, STREAM>;
    static EldAttrOnce once;
    { const int rc = once.ensure(kern, lds_bytes); if (rc) return rc; }
    ELD_LAUNCH(kern, dim3((unsigned)blocks), dim3(512), lds_bytes, st, a);
    ELD_LAUNCH_CHECK();
    return 0;
}

template <int WCO, int WCI, int WPIX, int TH, int TWT>
static int launch_w8d(WgradArgs a, hipStream_t st) {
    a.xcd = eld_xcd_mask() & XCD_WGRAD8;
    a.band = eld_tile_band();
    constexpr int COB = 32 * WCO, JBK = 32 * WCI;
    a.vp = vrow_pitch(a.N, a.H, TH);
    a.tiles_x = (a.W + TWT - 1) / TWT;
    a.tiles_y = (vrow_extent(a.N, a.H, a.vp) + TH - 1) / TH;
    constexpr int PIECES = WCO * (TH * TWT * 4 / 64) + WCI * (((TH + 2) * (TWT + 2) * 4 + 63) / 64);
    size_t lds_bytes = (size_t)2 * PIECES * 1024;
    const size_t red_bytes = (size_t)7 * 16 * 64 * sizeof(float);
    if (lds_bytes < red_bytes) lds_bytes = red_bytes;
    const long long blocks = (long long)(a.CA / COB) * (a.CBp / JBK) * a.psplit;
    if (blocks <= 0) return 0;
    auto kern = wgrad8d_kernel<WCO, WCI, WPIX, TH, TWT>;
    static EldAttrOnce once;
    { const int rc = once.ensure(kern, lds_bytes); if (rc) return rc; }
    ELD_LAUNCH(kern, dim3((unsigned)blocks), dim3(512), lds_bytes, st, a);
    ELD_LAUNCH_CHECK();
    return 0;
}

static int launch_wgrad8(const WgradArgs& a, hipStream_t st) {
    const bool bf16 = a.dtype == DT_BF16;
    const size_t es = bf16 ? 2 : 4;
    int COB, JBK, TH, TWo;
    if (!wgrad8_shape(a.CA, a.CBp, COB, JBK, TH, TWo, bf16)) return ELD_ENOTSUP;
    if ((a.C0 % 32) || (a.C1 % 32)) return ELD_ENOTSUP;
    // both operands are addressed through a window of two images (virtual rows): it must fit a buffer descriptor with 32-bit offsets
    const size_t win = a.N > 1 ? 2 : 1;
    if ((size_t)a.H * a.W * a.CA * es * win >= 0xFFFFFFF0ull || (size_t)a.H * a.W * (a.C0 > a.C1 ? a.C0 : a.C1) * es * win >= 0xFFFFFFF0ull) return ELD_ENOTSUP;
    if (bf16 && wgrad8_dma() && COB == 128) {
        // (measured on the same box, profiles/r04_ab_notes.md: waves 4-7 issuing their pieces behind their first k-step +1.3 %, 20 x 8 tiles +5.5 %)
        return launch_w8d<4, 2, 1, 16, 8>(a, st);
    }
    if (bf16) {
        if (COB == 128) return launch_w8<bf16_t, 4, 2, 1, 32, 8>(a, st);
        if (COB == 64 && JBK == 64) return launch_w8<bf16_t, 2, 2, 2, 8, 32>(a, st);
        if (COB == 64) return launch_w8<bf16_t, 2, 1, 4, 8, 32>(a, st);
        if (JBK == 64) return launch_w8<bf16_t, 1, 2, 4, 8, 32>(a, st);
        return launch_w8<bf16_t, 1, 1, 8, 8, 32>(a, st);
    }
    // round 6, opt-in (ELD_WG8_STREAM: bit 1 = the 128 x 64 blocks, bit 2 = the others): streamed fragment blocks in the main loop, as conv_x3.hip's.  Measured
    // same-box against the round-5 loop: 128 x 64 blocks 1997 -> 2005 us (+0.4 %), 64 x 64 2872 -> 2860 (-0.4 %), 32 x 32 2456 -> 2482 (+1.0 %): this loop's
    // eighteen-read bursts per kernel row were already covered by the SIMD's other wave.  Default: the round-5 loop.
    static const int stream = [] { const char* e = getenv("ELD_WG8_STREAM"); return e ? atoi(e) : 0; }();
    // ELD_WG8_ROWSHARE (opt-in): the 128 x 64 blocks pair a k-step's tile rows TH/2 apart and share X fragments between kernel rows (wgrad8_kernel, STREAM == 2).
    // Measured same-box (profiles/r06_ab_notes.md): 45 % fewer LDS fragment reads, 1979 -> 2008 us per launch (+1.5 %), step unchanged: this kernel is not held by
    // its LDS reads, and the units at the tile's first and last row pair feed only six MFMAs per fragment triple.
    static const int rowshare = [] { const char* e = getenv("ELD_WG8_ROWSHARE"); return e ? atoi(e) : 0; }();
    if (COB == 128 && rowshare) return launch_w8<float, 4, 2, 1, 8, 8, 2>(a, st);
    if (COB == 128) return (stream & 1) ? launch_w8<float, 4, 2, 1, 8, 8, 1>(a, st) : launch_w8<float, 4, 2, 1, 8, 8>(a, st);
    if (stream & 2) {
        if (COB == 64 && JBK == 64) return launch_w8<float, 2, 2, 2, 2, 32, 1>(a, st);
        if (COB == 64) return launch_w8<float, 2, 1, 4, 4, 32, 1>(a, st);
        if (JBK == 64) return launch_w8<float, 1, 2, 4, 4, 32, 1>(a, st);
        return launch_w8<float, 1, 1, 8, 4, 32, 1>(a, st);
    }
    if (COB == 64 && JBK == 64) return launch_w8<float, 2, 2, 2, 2, 32>(a, st);
    if (COB == 64) return launch_w8<float, 2, 1, 4, 4, 32>(a, st);
    if (JBK == 64) return launch_w8<float, 1, 2, 4, 4, 32>(a, st);
    return launch_w8<float, 1, 1, 8, 4, 32>(a, st);
}

template <typename T, int MODE, int WCO, int TH, int ALG = ALG_F32>
static int launch_w(WgradArgs a, hipStream_t st) {
    a.xcd = eld_xcd_mask() & XCD_WGRAD;
    constexpr int COB = 32 * WCO;
    constexpr int X_PIX = MODE == CONV_3X3 ? (TH + 2) * (TW + 2) : 4 * TH * TW;
    a.tiles_x = (a.W + TW - 1) / TW;
    a.tiles_y = (a.H + TH - 1) / TH;
    constexpr int GROWS = TH * TW + ((ALG == ALG_X3 || ALG == ALG_H2) ? 1 : 0);      // one pad row per 32-channel block of G in the fp32 piece layouts (see the kernel)
    size_t lds_bytes = (size_t)(GROWS * COB + X_PIX * JB) * (ALG == ALG_X3 ? 6 : (ALG == ALG_BFM ? 2 : 4));      // ALG_H2: 2 planes x 2 B
    const size_t red_bytes = (size_t)4 * 16 * 64 * sizeof(float);
    if (lds_bytes < red_bytes) lds_bytes = red_bytes;
    const long long blocks = (long long)(a.CA / COB) * (a.CBp / JB) * a.psplit;
    if (blocks <= 0) return 0;
    auto kern = wgrad_kernel<T, MODE, WCO, TH, ALG>;
    static EldAttrOnce once;
    { const int rc = once.ensure(kern, lds_bytes); if (rc) return rc; }
    ELD_LAUNCH(kern, dim3((unsigned)blocks), dim3(256), lds_bytes, st, a);
    ELD_LAUNCH_CHECK();
    return 0;
}

int launch_wgrad(const WgradArgs& a, int mode, hipStream_t st) {
    if (a.CA % 32 || a.CBp % 32 || a.C0 % 4 || a.C1 % 4 || a.psplit < 1) return ELD_EINVAL;
    if (a.C1 > 0 && a.C0 % 32) return ELD_EINVAL;
    const bool c64 = a.CA % 64 == 0;
    if (a.dtype == DT_BF16) {
        if (a.C0 % 8 || a.C1 % 8) return ELD_EINVAL;
        if (mode == CONV_3X3 && a.wgrad8) {                 // 8-wave re-blocked kernel (partials sized for its block shape)
            const int rc = launch_wgrad8(a, st);
            if (rc != ELD_ENOTSUP) return rc;
        }
        static const int mma = [] { const char* e = getenv("ELD_WGRAD_BF16_MMA"); return e ? atoi(e) : 1; }();      // ELD_WGRAD_BF16_MMA=0 falls back to fp32-MFMA accumulation of the widened operands
        if (mma) {
            if (mode == CONV_3X3) return c64 ? launch_w<bf16_t, CONV_3X3, 2, 4, ALG_BFM>(a, st) : launch_w<bf16_t, CONV_3X3, 1, 4, ALG_BFM>(a, st);
            if (mode == CONV_GATHER2X2) return c64 ? launch_w<bf16_t, CONV_GATHER2X2, 2, 2, ALG_BFM>(a, st) : launch_w<bf16_t, CONV_GATHER2X2, 1, 2, ALG_BFM>(a, st);
            return ELD_EINVAL;
        }
        if (mode == CONV_3X3) return c64 ? launch_w<bf16_t, CONV_3X3, 2, 4>(a, st) : launch_w<bf16_t, CONV_3X3, 1, 4>(a, st);
        if (mode == CONV_GATHER2X2) return c64 ? launch_w<bf16_t, CONV_GATHER2X2, 2, 2>(a, st) : launch_w<bf16_t, CONV_GATHER2X2, 1, 2>(a, st);
        return ELD_EINVAL;
    }
    const int algo = resolve_algo(a.algo);
    if (algo == 2) {
        if (!a.amax_g || !a.amax_x0) return ELD_EINVAL;
        // 64-channel blocks: 2-row tiles (the 4-row variant needs 60 staging registers on top of 144 accumulators and spills)
        if (mode == CONV_3X3) return c64 ? launch_w<float, CONV_3X3, 2, 2, ALG_H2>(a, st) : launch_w<float, CONV_3X3, 1, 4, ALG_H2>(a, st);
        if (mode == CONV_GATHER2X2) return c64 ? launch_w<float, CONV_GATHER2X2, 2, 2, ALG_H2>(a, st) : launch_w<float, CONV_GATHER2X2, 1, 2, ALG_H2>(a, st);
        return ELD_EINVAL;
    }
    if (mode == CONV_3X3 && algo == 1 && a.wgrad8) {
        const int rc = launch_wgrad8(a, st);
        if (rc != ELD_ENOTSUP) return rc;
    }
    if (mode == CONV_3X3 && algo == 1) return c64 ? launch_w<float, CONV_3X3, 2, 2, ALG_X3>(a, st) : launch_w<float, CONV_3X3, 1, 2, ALG_X3>(a, st);
    if (mode == CONV_3X3) return c64 ? launch_w<float, CONV_3X3, 2, 4>(a, st) : launch_w<float, CONV_3X3, 1, 4>(a, st);
    if (mode == CONV_GATHER2X2 && algo == 1 && a.wgrad8 && a.C1 == 0 && a.C0 == a.CBp && wgradt8_takes(a.CA, a.C0)) return launch_wt8(a, st);      // partials sized for the 128 x 64 blocks (unet.hip wgrad_geom)
    if (mode == CONV_GATHER2X2 && algo == 1) return c64 ? launch_w<float, CONV_GATHER2X2, 2, 2, ALG_X3>(a, st) : launch_w<float, CONV_GATHER2X2, 1, 2, ALG_X3>(a, st);
    if (mode == CONV_GATHER2X2) return c64 ? launch_w<float, CONV_GATHER2X2, 2, 2>(a, st) : launch_w<float, CONV_GATHER2X2, 1, 2>(a, st);
    return ELD_EINVAL;
}

// out[(i*CBr + j)*T + tap] = sum_s part[s][tap][i][j];  bgrad[i] = sum_s bpart[s][i].
// A workgroup = 64 consecutive (i, j) pairs (lanes: coalesced 256-byte reads along j) x NW waves; wave w sums the partials
// w, w + NW, ... for all T taps with 4 loads in flight, the waves are combined through LDS in a fixed order (run-to-run
// bit-stable), and wave 0 writes the pair's T consecutive outputs.  The last ceil(CA/64) workgroups do the bias sums the
// same way.
template <int T>
__global__ __launch_bounds__(1024) void wgrad_reduce_kernel(const float* __restrict__ part, const float* __restrict__ bpart, float* __restrict__ wgrad,
                                                            float* __restrict__ bgrad, int psplit, int CA, int CBp, int CBr, int pair_blocks, int NB) {
    extern __shared__ float sh[];                 // [NW][T][64]
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, NW = blockDim.x >> 6;
    const size_t pairs = (size_t)CA * CBp, plane = pairs * T;
    if ((int)blockIdx.x >= pair_blocks) {         // bias block: 64 of the NB bias channels (bpart is [psplit][NB])
        const int c = ((int)blockIdx.x - pair_blocks) * 64 + lane;
        float b = 0.f;
        if (c < NB) {
            int k = w;
            for (; k + 3 * NW < psplit; k += 4 * NW)
                b += (bpart[(size_t)k * NB + c] + bpart[(size_t)(k + NW) * NB + c]) + (bpart[(size_t)(k + 2 * NW) * NB + c] + bpart[(size_t)(k + 3 * NW) * NB + c]);
            for (; k < psplit; k += NW) b += bpart[(size_t)k * NB + c];
        }
        sh[w * 64 + lane] = b;
        __syncthreads();
        if (w == 0 && c < NB) {
            for (int ww = 1; ww < NW; ++ww) b += sh[ww * 64 + lane];
            bgrad[c] = b;
        }
        return;
    }
    const size_t pr = (size_t)blockIdx.x * 64 + lane;
    float s[T];
#pragma unroll
    for (int t = 0; t < T; ++t) s[t] = 0.f;
    if (pr < pairs) {
        int k = w;
        for (; k + NW < psplit; k += 2 * NW) {
            const float* p0 = part + (size_t)k * plane + pr;
            const float* p1 = part + (size_t)(k + NW) * plane + pr;
            float a[T], b[T];
#pragma unroll
            for (int t = 0; t < T; ++t) { a[t] = p0[(size_t)t * pairs]; b[t] = p1[(size_t)t * pairs]; }
#pragma unroll
            for (int t = 0; t < T; ++t) s[t] += a[t] + b[t];
        }
        for (; k < psplit; k += NW) {
            const float* p0 = part + (size_t)k * plane + pr;
#pragma unroll
            for (int t = 0; t < T; ++t) s[t] += p0[(size_t)t * pairs];
        }
    }
    if (NW > 1) {
#pragma unroll
        for (int t = 0; t < T; ++t) sh[(w * T + t) * 64 + lane] = s[t];
        __syncthreads();
        if (w == 0)
            for (int ww = 1; ww < NW; ++ww)
#pragma unroll
                for (int t = 0; t < T; ++t) s[t] += sh[(ww * T + t) * 64 + lane];
    }
    if (w == 0 && pr < pairs) {
        const int j = (int)(pr % CBp), i = (int)(pr / CBp);
        if (j < CBr) {
            float* o = wgrad + ((size_t)i * CBr + j) * T;
#pragma unroll
            for (int t = 0; t < T; ++t) o[t] = s[t];
        }
    }
}

int launch_wgrad_reduce(const float* part, const float* bpart, float* wgrad, float* bgrad, int psplit, int T, int CA,
                        int CBp, int CBr, hipStream_t st, int bias_n) {
    const int NB = bias_n > 0 ? bias_n : CA;      // channels of the bias partials (transposed convs: the column sums of the gathered operand, CBp of them)
    if (T != 9 && T != 4) return ELD_EINVAL;
    int NW = 1;
    while (NW < 16 && NW < psplit) NW <<= 1;
    const size_t pairs = (size_t)CA * CBp;
    const int pair_blocks = (int)((pairs + 63) / 64);
    const int blocks = pair_blocks + (bgrad ? (NB + 63) / 64 : 0);
    const size_t lds = (size_t)NW * T * 64 * sizeof(float);
    if (T == 9) ELD_LAUNCH((wgrad_reduce_kernel<9>), dim3(blocks), dim3(64 * NW), lds, st, part, bpart, wgrad, bgrad, psplit, CA, CBp, CBr, pair_blocks, NB);
    else ELD_LAUNCH((wgrad_reduce_kernel<4>), dim3(blocks), dim3(64 * NW), lds, st, part, bpart, wgrad, bgrad, psplit, CA, CBp, CBr, pair_blocks, NB);
    ELD_LAUNCH_CHECK();
    return 0;
}
