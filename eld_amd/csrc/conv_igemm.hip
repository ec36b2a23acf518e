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
#include <stdlib.h>
#include <atomic>
#include "conv.h"

#define ELD_FP32_CONV_DEFAULT 1
#define PS 20           // LDS pixel stride in 4-byte words: one 64-byte K chunk (16 fp32 / 32 bf16 channels) + 16 bytes pad
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
#define TW 32

// experimental (ELD_FP32_CONV=h2): four fp32 values * scale -> two fp16 pieces each (round-toward-zero, residual exact), piece p
// at byte offset 32p of the LDS row
__device__ __forceinline__ void split_h2(float* row, float4 v, float scale) {
    typedef __fp16 h2 __attribute__((ext_vector_type(2)));
    const float x0 = v.x * scale, x1 = v.y * scale, x2 = v.z * scale, x3 = v.w * scale;
    const h2 a01 = __builtin_amdgcn_cvt_pkrtz(x0, x1), a23 = __builtin_amdgcn_cvt_pkrtz(x2, x3);
    // first piece truncated (the residual is then exact and has the operand's sign), second piece rounded to nearest: the
    // pair represents the scaled operand to 2^-22 relative, without the bias two truncations would leave
    const h2 b01 = {(__fp16)(x0 - (float)a01[0]), (__fp16)(x1 - (float)a01[1])}, b23 = {(__fp16)(x2 - (float)a23[0]), (__fp16)(x3 - (float)a23[1])};
    *reinterpret_cast<uint2*>(row) = make_uint2(__builtin_bit_cast(unsigned, a01), __builtin_bit_cast(unsigned, a23));
    *reinterpret_cast<uint2*>(row + 8) = make_uint2(__builtin_bit_cast(unsigned, b01), __builtin_bit_cast(unsigned, b23));
}

template <int MODE, int RPW>
struct Geo {
    static constexpr int TH = 4 * RPW;
    static constexpr int TAPS = MODE == CONV_3X3 ? 9 : (MODE == CONV_1X1 ? 1 : 4);
    static constexpr int A_PIX = MODE == CONV_3X3 ? (TH + 2) * (TW + 2) : (MODE == CONV_1X1 ? TH * TW : 4 * TH * TW);
};

template <typename T, int MODE, int BN, int RPW, bool H2 = false>
__global__ __launch_bounds__(256, 2) void conv_igemm_kernel(const ConvArgs a) {
    using G = Geo<MODE, RPW>;
    constexpr int ES = sizeof(T);               // element size of activations / packed weights
    constexpr int CK = 64 / ES;                 // channels per K chunk: 16 (fp32) or 32 (bf16) -- always 64 bytes per pixel
    constexpr int TH = G::TH, TAPS = G::TAPS, A_PIX = G::A_PIX, NT = BN / 32;
    constexpr int A_WORDS = A_PIX * PS, B_ROWS = TAPS * BN;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* ldsA = lds;
    float* ldsB = lds + A_WORDS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = lane & 31, hi = lane >> 5;
    const int NB = a.Nout / BN;
    const int Cin = a.C0 + a.C1;
    const int Hs = MODE == CONV_GATHER2X2 ? 2 * a.H : a.H;     // source dims
    const int Ws = MODE == CONV_GATHER2X2 ? 2 * a.W : a.W;
    const int total_tiles = a.tiles_x * a.tiles_y * a.N * NB;
    const int Cs0 = a.C0;                                       // channels per source tensor
    float sc_a = 1.0f, sc_w = 1.0f;                              // H2: power-of-two operand scales (exact), undone in the epilogue
    if constexpr (H2) {
        float ma = *a.amax_in0;
        if (a.amax_in1) ma = fmaxf(ma, *a.amax_in1);
        sc_a = h2_scale(ma);
        sc_w = h2_scale(*a.amax_w);
    }

    // Persistent workgroup: walks tiles blockIdx.x, +gridDim.x, ... and streams (tile, chunk) work items
    // through a register-staged software pipeline (write-after-barrier): the global loads of the NEXT work
    // item -- the next K chunk, or the first chunk of the next tile -- are issued before the MFMA phase of
    // the current one and land in registers while the matrix pipe works; they go to LDS after the barrier
    // that retires the current reads.  The pipeline therefore never drains between tiles; only the
    // ds_write pass, two barriers per chunk and the epilogue stores are not covered by MFMA work.
    constexpr int A_UNITS = A_PIX * 4, B_UNITS = B_ROWS * 4;
    constexpr int A_IT = (A_UNITS + 255) / 256, B_IT = (B_UNITS + 255) / 256;
    float4 ra[A_IT], rb[B_IT];
    // Staging loads are buffer loads: one 32-bit byte offset per load (instead of a 64-bit address pair),
    // the descriptor in SGPRs, the per-chunk channel offset in the scalar offset, and hardware range
    // checking returns zeros for the OOB marker offset -> the conv's zero padding costs no predicate.
    constexpr unsigned OOB = 0xFFFFFFF0u;
    unsigned a_voff[A_IT];       // load side: byte offset of the staging unit inside the image, OOB = zero fill
    unsigned b_voff[B_IT];       // byte offset of the weight unit inside the n-block's slab (chunk offset is scalar)
    int l_nb = 0, l_img = 0;     // load side: tile being loaded (workgroup-uniform -> SGPRs)
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc((void*)a.wp, 0, (int)((size_t)TAPS * a.Nout * Cin * ES), 0x00020000);
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
        const int u = tid + it * 256;
        const int row = u >> 2, part = u & 3;
        const int tap = row / BN, n = row - tap * BN;
        b_voff[it] = u < B_UNITS ? (unsigned)((tap * a.Nout + n) * Cin * ES + part * 16) : OOB;
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
        int y0, x0;
        decode(t, l_nb, l_img, y0, x0);
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            const int u = tid + it * 256;
            const int hp = u >> 2, part = u & 3;
            int gy = 0, gx = 0;
            bool ok = u < A_UNITS;
            if (MODE == CONV_3X3) {
                const int hy = hp / (TW + 2), hx = hp - hy * (TW + 2);
                gy = y0 + hy - 1; gx = x0 + hx - 1;
                ok = ok && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
            } else if (MODE == CONV_1X1) {
                const int py = hp / TW, px = hp - py * TW;
                gy = y0 + py; gx = x0 + px;
                ok = ok && gy < a.H && gx < a.W;
            } else {
                const int tap = hp / (TH * TW), lp = hp - tap * (TH * TW);
                const int py = lp / TW, px = lp - py * TW;
                ok = ok && (y0 + py) < a.H && (x0 + px) < a.W;
                gy = 2 * (y0 + py) + (tap >> 1); gx = 2 * (x0 + px) + (tap & 1);
            }
            a_voff[it] = ok ? (unsigned)(gy * Ws + gx) * (unsigned)(Cs0 * ES) + (unsigned)part * 16u : OOB;     // final byte offset
        }
    };
    auto load_chunk = [&](int c0) {
        // both concat sources have the same channel count (checked at launch), so one set of offsets serves both
        const char* src = static_cast<const char*>(c0 < a.C0 ? a.in0 : a.in1);
        const int cs = c0 < a.C0 ? c0 : c0 - a.C0;
        const size_t img_bytes = (size_t)Hs * Ws * Cs0 * ES;
        const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc((void*)(src + (size_t)l_img * img_bytes), 0, (int)img_bytes, 0x00020000);
#pragma unroll
        for (int it = 0; it < A_IT; ++it)
            ra[it] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, (int)a_voff[it], cs * ES, 0));
        const int wsoff = (l_nb * BN * Cin + c0) * ES;
#pragma unroll
        for (int it = 0; it < B_IT; ++it)
            rb[it] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, (int)b_voff[it], wsoff, 0));
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            const int u = tid + it * 256;
            if constexpr (H2) { if (u < A_UNITS) split_h2(ldsA + (u >> 2) * PS + (u & 3) * 2, ra[it], sc_a); }
            else if (u < A_UNITS) *reinterpret_cast<float4*>(ldsA + (u >> 2) * PS + (u & 3) * 4) = ra[it];
        }
#pragma unroll
        for (int it = 0; it < B_IT; ++it) {
            const int u = tid + it * 256;
            if constexpr (H2) { if (u < B_UNITS) split_h2(ldsB + (u >> 2) * PS + (u & 3) * 2, rb[it], sc_w); }
            else if (u < B_UNITS) *reinterpret_cast<float4*>(ldsB + (u >> 2) * PS + (u & 3) * 4) = rb[it];
        }
    };

    int t = xcd_block(a.xcd);
    if (t >= total_tiles) return;
    setup_load(t);
    load_chunk(0);
    for (;;) {
        int nb, img, y0, x0;             // compute / epilogue side of the current tile
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
            __syncthreads();                 // every wave is done reading the previous chunk
            store_chunk();
            __syncthreads();
            if (!(ELD_DBG(a) & 4)) {
                if (c0 + CK < Cin) {
                    load_chunk(c0 + CK);         // in flight during the MFMA phase below
                } else if (t_next < total_tiles) {
                    setup_load(t_next);
                    load_chunk(0);
                }
            }
            if (ELD_DBG(a) & 2) continue;
            // ---- MFMA over taps x the chunk's 64 bytes of K.  Group g = (tap, q): every lane reads 16 bytes per operand
            //      row (fp32: 4 channels of its k-half -> 4 x v_mfma_f32_32x32x2_f32; bf16: 8 channels = one whole
            //      v_mfma_f32_32x32x16_bf16 operand).  Fragments of group g+1 are read from LDS while group g's MFMAs
            //      occupy the matrix pipe (explicit register double-buffering, order pinned with sched_barrier) -----------
            float4 fa[2][RPW], fb[2][NT];
            auto read_group = [&](int g, float4 (&A)[RPW], float4 (&B)[NT]) {
                const int tap = g >> 1, q = g & 1;
                const int ko = (ES == 4 && !H2) ? hi * 8 + q * 4 : hi * 4 + q * 8;      // H2: q = piece, bf16: q = k-step      // word offset of this lane's 16 bytes inside the pixel row
#pragma unroll
                for (int r = 0; r < RPW; ++r) {
                    const int row = wave * RPW + r;
                    int off;
                    if (MODE == CONV_3X3) off = ((row + tap / 3) * (TW + 2) + m + tap % 3) * PS;
                    else if (MODE == CONV_1X1) off = (row * TW + m) * PS;
                    else off = (tap * TH * TW + row * TW + m) * PS;
                    A[r] = *reinterpret_cast<const float4*>(ldsA + off + ko);
                }
#pragma unroll
                for (int tt = 0; tt < NT; ++tt) B[tt] = *reinterpret_cast<const float4*>(ldsB + (tap * BN + tt * 32 + m) * PS + ko);
            };
            if constexpr (H2) {
                // two fp16 pieces per operand (x = x1 + x2 to 22 bits): x1 w1 + x1 w2 + x2 w1 on v_mfma_f32_32x32x16_f16
#pragma unroll
                for (int tap = 0; tap < TAPS; ++tap) {
                    read_group(2 * tap, fa[0], fb[0]);
                    read_group(2 * tap + 1, fa[1], fb[1]);
#pragma unroll
                    for (int r = 0; r < RPW; ++r)
#pragma unroll
                        for (int tt = 0; tt < NT; ++tt) {
                            acc[r][tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, fb[1][tt]), __builtin_bit_cast(half8, fa[0][r]), acc[r][tt], 0, 0, 0);
                            acc[r][tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, fb[0][tt]), __builtin_bit_cast(half8, fa[1][r]), acc[r][tt], 0, 0, 0);
                            acc[r][tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, fb[0][tt]), __builtin_bit_cast(half8, fa[0][r]), acc[r][tt], 0, 0, 0);
                        }
                    if constexpr (RPW * NT >= 4 && RPW == 4) __builtin_amdgcn_sched_barrier(0);      // 16-row tiles: keep later taps' reads from piling up (register budget)
                }
            } else {
            read_group(0, fa[0], fb[0]);
#pragma unroll
            for (int g = 0; g < 2 * TAPS; ++g) {
                const int cur = g & 1;
                if (g + 1 < 2 * TAPS) read_group(g + 1, fa[cur ^ 1], fb[cur ^ 1]);
                __builtin_amdgcn_sched_barrier(0);      // keep the prefetch reads ABOVE this group's MFMAs
                if constexpr (ES == 4) {
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
                        for (int r = 0; r < RPW; ++r) {
                            const float af = kk == 0 ? fa[cur][r].x : kk == 1 ? fa[cur][r].y : kk == 2 ? fa[cur][r].z : fa[cur][r].w;
#pragma unroll
                            for (int tt = 0; tt < NT; ++tt) {
                                const float bf = kk == 0 ? fb[cur][tt].x : kk == 1 ? fb[cur][tt].y : kk == 2 ? fb[cur][tt].z : fb[cur][tt].w;
                                acc[r][tt] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf, af, acc[r][tt], 0, 0, 0);      // D[channel][pixel]
                            }
                        }
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < RPW; ++r)
#pragma unroll
                        for (int tt = 0; tt < NT; ++tt)
                            acc[r][tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fb[cur][tt]), __builtin_bit_cast(bf16x8, fa[cur][r]),
                                                                                 acc[r][tt], 0, 0, 0);      // D[channel][pixel]
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            }
        }

        // ---- epilogue.  The MFMA ran as D[channel][pixel] (A = weights, B = pixels), so lane (m, hi) owns pixel
        //      x0 + m and, in accumulator quad q, the four CONSECUTIVE channels 8q + 4hi .. +3 of its 32-channel block:
        //      every access is one 16-byte dwordx4 per lane (4 per 32x32 tile), the bounds test is one lane mask, and
        //      all loads (bias, saved activations) are issued before the first store.
        float tmax0 = 0.f, tmax1 = 0.f;     // H2: max|out| of this tile per destination tensor
        {
            const int x = x0 + m;
            const bool xok = x < a.W;
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                const int y = y0 + wave * RPW + r;
                if (y >= a.H || (ELD_DBG(a) & 1) || !xok) continue;
                const size_t pix = (size_t)(img * a.H + y) * a.W + x;
#pragma unroll
                for (int tt = 0; tt < NT; ++tt) {
                    const int nbase = nb * BN + tt * 32 + 4 * hi;          // + 8q
                    float4 v[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = make_float4(acc[r][tt][4 * q], acc[r][tt][4 * q + 1], acc[r][tt][4 * q + 2], acc[r][tt][4 * q + 3]);
                    if constexpr (H2) {
                        const float ia = 1.0f / sc_a, iw = 1.0f / sc_w;      // powers of two: exact
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            v[q].x = v[q].x * ia * iw; v[q].y = v[q].y * ia * iw; v[q].z = v[q].z * ia * iw; v[q].w = v[q].w * ia * iw;
                        }
                    }
                    T* dst[4];
                    if (a.epi == EPI_FWD) {
                        float4 bs[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) bs[q] = *reinterpret_cast<const float4*>(a.bias + nbase + 8 * q);
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            v[q].x += bs[q].x; v[q].y += bs[q].y; v[q].z += bs[q].z; v[q].w += bs[q].w;
                            if (a.lrelu) {
                                v[q].x = fmaxf(0.2f * v[q].x, v[q].x); v[q].y = fmaxf(0.2f * v[q].y, v[q].y);
                                v[q].z = fmaxf(0.2f * v[q].z, v[q].z); v[q].w = fmaxf(0.2f * v[q].w, v[q].w);
                            }
                            dst[q] = static_cast<T*>(a.out0) + pix * a.Nout + nbase + 8 * q;
                        }
                    } else if (a.epi == EPI_CONVT_FWD) {
                        float4 bs[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int n = nbase + 8 * q;
                            const int tap = n / a.Cout_t, co = n - tap * a.Cout_t;
                            bs[q] = *reinterpret_cast<const float4*>(a.bias + co);
                            dst[q] = static_cast<T*>(a.out0) + ((size_t)(img * 2 * a.H + 2 * y + (tap >> 1)) * (2 * a.W) + 2 * x + (tap & 1)) * a.Cout_t + co;
                        }
#pragma unroll
                        for (int q = 0; q < 4; ++q) { v[q].x += bs[q].x; v[q].y += bs[q].y; v[q].z += bs[q].z; v[q].w += bs[q].w; }
                    } else {
                        float4 s[4];
                        bool has[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int n = nbase + 8 * q;
                            const bool lo = n < a.split;
                            const int C = lo ? a.split : a.Nout - a.split;
                            const size_t idx = pix * C + (lo ? n : n - a.split);
                            dst[q] = static_cast<T*>(lo ? a.out0 : a.out1) + idx;
                            const T* act = static_cast<const T*>(lo ? a.act0 : a.act1);
                            has[q] = act != nullptr;
                            s[q] = make_float4(1.f, 1.f, 1.f, 1.f);
                            if (has[q]) {
                                if constexpr (ES == 4) s[q] = *reinterpret_cast<const float4*>(act + idx);
                                else s[q] = unpack_bf4(*reinterpret_cast<const uint2*>(act + idx));
                            }
                        }
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            if (has[q]) {
                                v[q].x *= lrelu_slope(s[q].x); v[q].y *= lrelu_slope(s[q].y);
                                v[q].z *= lrelu_slope(s[q].z); v[q].w *= lrelu_slope(s[q].w);
                            }
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        asm volatile("" : "+v"(v[q].x), "+v"(v[q].y), "+v"(v[q].z), "+v"(v[q].w));     // final values: no load result is consumed below
                    }
                    if constexpr (H2) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float mq = fmaxf(fmaxf(fabsf(v[q].x), fabsf(v[q].y)), fmaxf(fabsf(v[q].z), fabsf(v[q].w)));
                            if (a.epi == EPI_GRAD && nbase + 8 * q >= a.split) tmax1 = fmaxf(tmax1, mq); else tmax0 = fmaxf(tmax0, mq);
                        }
                    }
                    if constexpr (ES == 4) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) *reinterpret_cast<float4*>(dst[q]) = v[q];
                    } else {
                        // bf16: 16-byte stores (conv.h bf16_pair_swap) -- after the exchange the lane owns the whole 8-channel group 2j + hi,
                        // which starts at dst[2j] + 4 hi (dst[q] = block base + 8 q + 4 hi in every epilogue mode: a 32-channel block never
                        // straddles the concat split or a transposed-conv tap)
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const uint4 w = bf16_pair_swap(pack_bf4(v[2 * j]), pack_bf4(v[2 * j + 1]));
                            *reinterpret_cast<uint4*>(dst[2 * j] + 4 * hi) = w;
                        }
                    }
                }
            }
        }
        if constexpr (H2) {                 // wave-uniform control flow: every lane takes part in the shuffles
            if (a.amax_out0) amax_accumulate(a.amax_out0, tmax0);
            if (a.amax_out1) amax_accumulate(a.amax_out1, tmax1);
        }
        if (t_next >= total_tiles) break;
        t = t_next;
    }
}


template <typename T, int MODE, int BN, int RPW, bool H2 = false>
static int launch_t(ConvArgs a, hipStream_t st) {
    a.xcd = eld_xcd_mask() & XCD_IGEMM;
    using G = Geo<MODE, RPW>;
    a.tiles_x = (a.W + TW - 1) / TW;
    a.tiles_y = (a.H + G::TH - 1) / G::TH;
    const size_t lds_bytes = (size_t)(G::A_PIX + G::TAPS * BN) * PS * sizeof(float);
    const long long tiles = (long long)a.tiles_x * a.tiles_y * a.N * (a.Nout / BN);
    if (tiles <= 0) return 0;
    if (tiles > 0x7fffffffLL) return ELD_ENOTSUP;
    auto kern = conv_igemm_kernel<T, MODE, BN, RPW, H2>;
    static EldAttrOnce once;          // per instantiation, per device
    { const int rc = once.ensure(kern, lds_bytes); if (rc) return rc; }
    // persistent grid: as many workgroups as are co-resident (LDS: 160 KiB per CU; registers: 2 waves per SIMD)
    int per_cu = (int)((160 * 1024) / lds_bytes);
    if (per_cu > 2) per_cu = 2;
    if (per_cu < 1) per_cu = 1;
    long long grid = (long long)eld_num_cus() * per_cu;
    if (grid > tiles) grid = tiles;
    ELD_LAUNCH(kern, dim3((unsigned)grid), dim3(256), lds_bytes, st, a);
    ELD_LAUNCH_CHECK();
    return 0;
}

template <typename T>
static int launch_dt(const ConvArgs& a, int mode, hipStream_t st) {
    const bool n64 = a.Nout % 64 == 0;
    switch (mode) {
        case CONV_3X3: return n64 ? launch_t<T, CONV_3X3, 64, 2>(a, st) : launch_t<T, CONV_3X3, 32, 2>(a, st);
        case CONV_1X1: return n64 ? launch_t<T, CONV_1X1, 64, 2>(a, st) : launch_t<T, CONV_1X1, 32, 2>(a, st);
        case CONV_GATHER2X2: return n64 ? launch_t<T, CONV_GATHER2X2, 64, 1>(a, st) : launch_t<T, CONV_GATHER2X2, 32, 1>(a, st);
    }
    return ELD_EINVAL;
}

int conv_fp32_algo(int set) {
    // the process default: initialised once (thread-safe magic static), then an atomic that concurrent callers may read and set
    static std::atomic<int> algo([] {
        const char* e = getenv("ELD_FP32_CONV");
        int a = (e && (e[0] == 'm' || e[0] == '0')) ? 0 : ELD_FP32_CONV_DEFAULT;
        if (e && (e[0] == 'x' || e[0] == '1')) a = 1;
        if (e && (e[0] == 'h' || e[0] == '2')) a = 2;
        return a;
    }());
    if (set >= 0 && set <= 2) return algo.exchange(set);
    return algo.load();
}

// ---- tile shape of the 3x3 kernels whose pixel slots are mapped per lane (conv_x3d_kernel, conv_bfd_kernel): see conv.h ----------------------
static long long tile_count(int N, int H, int W, int th, int tw) {
    const int vp = vrow_pitch(N, H, th);
    return (long long)((W + tw - 1) / tw) * ((vrow_extent(N, H, vp) + th - 1) / th);
}
// LDS cycles of one activation-fragment ds_read_b128 of a th x tw tile under the grouped slot numbering, relative to conflict-free (1.0): the halo
// pixels a lane group touches must be distinct modulo 16 (conv_x3d: 112-byte pixel rows, 28 P mod 64 words; conv_bfd: 64-byte rows with the octet
// XOR -- the same condition); every pair of a group that collides costs the group one more cycle.  Averaged over the waves / rows of a workgroup.
static float tile_conflict_factor(int th, int tw, int TH) {
    const int cols = TH, hw = tw + 2, npx = th * tw;             // TH columns of 32 slots per workgroup (waves x rows per wave)
    long long cyc = 0, n = 0;
    for (int col = 0; col < cols; ++col)
        for (int grp = 0; grp < 2; ++grp) {
            int cntm[16] = {0};
            for (int s = 0; s < 16; ++s) {
                int p = col * 32 + grp * 16 + s;
                if (p >= npx) p = 0;
                const int tr = p / tw, tc = p - tr * tw;
                ++cntm[(tr * hw + tc) & 15];
            }
            int mx = 1;
            for (int i = 0; i < 16; ++i) mx = cntm[i] > mx ? cntm[i] : mx;
            cyc += mx; ++n;
        }
    return (float)cyc / (float)n;
}
void conv_tile_shape(int N, int H, int W, int TH, bool pooled, int& th, int& tw) {
    th = TH; tw = 32;
    // ELD_CONV_TILES (A/B runs of the tile choice; virtual rows stay): f = TH x 32 everywhere, p = widths 8/16/32/64 only, m<k> = widths >= k,
    // q<k> = widths that are multiples of k, t = fewest tiles (no conflict term)
    // (function-local statics with initialisers: C++11 guarantees one thread-safe initialisation -- the entry points may be called from several host threads)
    struct TileMode { int mode, marg; };
    static const TileMode tm = [] {
        TileMode t = {0, 0};
        const char* e = getenv("ELD_CONV_TILES");
        if (e && e[0] == 'f') t.mode = 1;
        if (e && e[0] == 'p') t.mode = 2;
        if (e && e[0] == 'm') { t.mode = 3; t.marg = atoi(e + 1); }
        if (e && e[0] == 'q') { t.mode = 4; t.marg = atoi(e + 1) > 0 ? atoi(e + 1) : 1; }
        if (e && e[0] == 't') t.mode = 5;
        return t;
    }();
    const int mode = tm.mode, marg = tm.marg;
    if (pooled || mode == 1 || N <= 0 || H <= 0 || W <= 0 || (TH != 8 && TH != 16 && TH != 32)) return;
    const int ti = TH == 8 ? 0 : (TH == 16 ? 1 : 2);
    const int slots = TH * 32, halo = (TH + 2) * 34;
    auto height = [&](int w) { int h = slots / w; while (h > 1 && (h + 2) * (w + 2) > halo) --h; return h > 128 ? 128 : h; };
    struct ConflictTable { float v[3][65]; };                // conflict factor per width, per TH in {8, 16, 32}; filled once (shape-independent)
    static const ConflictTable cft = [] {
        ConflictTable c = {};
        for (int k = 0; k < 3; ++k) {
            const int THk = 8 << k, sl = THk * 32, hl = (THk + 2) * 34;
            for (int w = 8; w <= 64; ++w) {
                int h = sl / w;
                while (h > 1 && (h + 2) * (w + 2) > hl) --h;
                c.v[k][w] = tile_conflict_factor(h > 128 ? 128 : h, w, THk);
            }
        }
        return c;
    }();
    const float (&cf)[3][65] = cft.v;
    double best = (double)tile_count(N, H, W, th, tw);       // the standard shape is conflict-free
    for (int w = 8; w <= 64; ++w) {                          // narrower than 8 pixels: a halo row is no longer a few whole 64-byte DMA units
        if (mode == 2 && (w & (w - 1))) continue;
        if (mode == 3 && w < marg) continue;
        if (mode == 4 && w % marg) continue;
        const int h = height(w);
        if (h == TH && w == 32) continue;
        const double c = (double)tile_count(N, H, W, h, w) * (mode == 5 ? 1.0 : 1.0 + 0.01 * ((double)cf[ti][w] - 1.0));
        if (c < best) { best = c; th = h; tw = w; }
    }
}
long long conv_tile_count(int N, int H, int W, int TH, bool pooled) {
    int th, tw;
    conv_tile_shape(N, H, W, TH, pooled, th, tw);
    return tile_count(N, H, W, th, tw);
}

int launch_conv(const ConvArgs& a_in, int mode, hipStream_t st) {
    ConvArgs a = a_in;
    a.dbg = 0;
#if ELD_DEV_TOOLS
    { static const int dbg = [] { const char* e = getenv("ELD_CONV_DBG"); return e ? atoi(e) : 0; }(); a.dbg = dbg; }
#endif
    const int Cin = a.C0 + a.C1;
    const int ck = a.dtype == DT_BF16 ? 32 : 16;
    if (Cin % ck || a.C0 % ck || a.Nout % 32) return ELD_EINVAL;
    if (a.C1 != 0 && a.C1 != a.C0) return ELD_ENOTSUP;          // virtual concat of two equally wide tensors (all the U-Net needs)
    if (a.epi == EPI_GRAD && (a.split % 32)) return ELD_EINVAL;
    if (a.epi == EPI_CONVT_FWD && (a.Cout_t % 4)) return ELD_EINVAL;
    if (a.dtype == DT_BF16) {
        if (mode == CONV_3X3 && a.epi != EPI_CONVT_FWD && bfd_slab_bn(a.Nout, Cin, a.N, a.H, a.W)) return launch_conv_bfd(a, st);      // weights in slab layout
        if (mode != CONV_3X3 && !a.pool_out && bfg_slab_bn(mode == CONV_GATHER2X2, a.Nout, a.C0, a.Cout_t, a.N, a.H, a.W)) return launch_conv_bfg(a, mode, st);      // weights in bfg slab layout
        return a.pool_out ? ELD_ENOTSUP : launch_dt<bf16_t>(a, mode, st);
    }
    const int algo = resolve_algo(a.algo);
    if (a.pool_out && !(mode == CONV_3X3 && a.epi == EPI_FWD && algo == 1 && a.dtype == DT_F32)) return ELD_ENOTSUP;     // fused pooling: conv_x3.hip only
    if (mode == CONV_3X3 && a.epi != EPI_CONVT_FWD && algo == 1) return launch_conv_x3(a, st);
    if (mode != CONV_3X3 && algo == 1) {
        const int rc = launch_conv_x3_gemm(a, mode, st);
        if (rc != ELD_ENOTSUP) return rc;
    }
    if (algo == 2) {                    // two fp16 pieces per operand, three products; needs the operand bounds
        if (!a.amax_in0 || !a.amax_w) return ELD_EINVAL;
        const bool n64 = a.Nout % 64 == 0;
        switch (mode) {
            case CONV_3X3: return n64 ? launch_t<float, CONV_3X3, 64, 2, true>(a, st) : launch_t<float, CONV_3X3, 32, 4, true>(a, st);
            case CONV_1X1: return n64 ? launch_t<float, CONV_1X1, 64, 2, true>(a, st) : launch_t<float, CONV_1X1, 32, 2, true>(a, st);
            case CONV_GATHER2X2: return n64 ? launch_t<float, CONV_GATHER2X2, 64, 1, true>(a, st) : launch_t<float, CONV_GATHER2X2, 32, 1, true>(a, st);
        }
        return ELD_EINVAL;
    }
    return launch_dt<float>(a, mode, st);
}
