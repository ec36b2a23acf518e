// conv_x3_dev.h -- device helpers shared by the three-piece fp32 3x3 kernels (conv_x3.hip, conv_x3w.hip): the exact three-piece cut of four fp32 values
// into an LDS row, the staging-row permutation, the fused 2x2 max-pool epilogue, the LDS-DMA piece, the streamed fragment blocks of a stage.
#pragma once
#include "conv.h"

namespace {

#define PX 28
#define TW 32
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));


__device__ __forceinline__ unsigned hi_pair(unsigned lo_src, unsigned hi_src) {        // (hi_src & 0xFFFF0000) | (lo_src >> 16)
    return __builtin_amdgcn_perm(hi_src, lo_src, 0x07060302u);
}

// cut four fp32 values into their three bf16 pieces and write piece p at word offset 8p of the LDS row
__device__ __forceinline__ void split_store(float* row, float4 v) {
    const unsigned x0 = __float_as_uint(v.x), x1 = __float_as_uint(v.y), x2 = __float_as_uint(v.z), x3 = __float_as_uint(v.w);
    const float r0 = v.x - __uint_as_float(x0 & 0xFFFF0000u), r1 = v.y - __uint_as_float(x1 & 0xFFFF0000u);
    const float r2 = v.z - __uint_as_float(x2 & 0xFFFF0000u), r3 = v.w - __uint_as_float(x3 & 0xFFFF0000u);
    const unsigned y0 = __float_as_uint(r0), y1 = __float_as_uint(r1), y2 = __float_as_uint(r2), y3 = __float_as_uint(r3);
    const float s0 = r0 - __uint_as_float(y0 & 0xFFFF0000u), s1 = r1 - __uint_as_float(y1 & 0xFFFF0000u);
    const float s2 = r2 - __uint_as_float(y2 & 0xFFFF0000u), s3 = r3 - __uint_as_float(y3 & 0xFFFF0000u);
    *reinterpret_cast<uint2*>(row) = make_uint2(hi_pair(x0, x1), hi_pair(x2, x3));
    *reinterpret_cast<uint2*>(row + 8) = make_uint2(hi_pair(y0, y1), hi_pair(y2, y3));
    *reinterpret_cast<uint2*>(row + 16) = make_uint2(hi_pair(__float_as_uint(s0), __float_as_uint(s1)), hi_pair(__float_as_uint(s2), __float_as_uint(s3)));
}


// Which LDS row a staging unit's 4 lanes fill.  ds_write_b64 is serviced in groups of 16 consecutive lanes (32 banks of 4 B): rows r .. r+3 at the
// 28-word row stride start at banks 0, 28, 24, 20 and their 8-word piece windows overlap pairwise (2-way conflicts on half the banks, measured as
// 27-100 % of the LDS-active cycles of these kernels); rows r, r+2, r+4, r+6 start at banks 0, 24, 16, 8: disjoint.  So inside every aligned
// block of 8 rows the staging order is 0,2,4,6,1,3,5,7 -- a bijection on [0, limit) (a trailing partial block keeps its order); loads and
// stores use the same map, the LDS image is unchanged.
__device__ __forceinline__ int stage_row(int r, int limit) {
    const int b = r & ~7, j = r & 7;
    return b + 8 > limit ? r : b + ((j & 3) << 1) + (j >> 2);
}

// Fused nn.MaxPool2d(2) of a forward tile (models/arch/Unet.py:51-63): a lane owns pixel column x0+m of RPW consecutive rows starting at an even
// row, so the vertical pair is in its own registers and the horizontal one in lane ^ 1 (conv.h fmax_lane_xor1).  acc holds the FINISHED values
// (the epilogue adds bias and applies LeakyReLU in place before it stores), so this pools exactly what was stored; the even lanes write
// [N, H/2, W/2, Nout].  H and W are even (checked by the launcher), so a window is never split by the image border.  Every lane of the wave runs
// the exchange; only the stores are predicated.
template <int RPW, int NT, int BN, bool PCODES = (NT <= 2)>
__device__ __forceinline__ void pool_epilogue(const ConvArgs& a, const f32x16 (&acc)[RPW][NT], const int (&img_p)[RPW / 2], int nb,
                                              const int (&y_p)[RPW / 2] /* first row of each of the lane's row pairs; >= H: none */, int x, int hi) {
    static_assert(RPW % 2 == 0, "row pairs per lane");
    const int Hp = a.H >> 1, Wp = a.W >> 1;
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) {
        const int nbase = nb * BN + tt * 32 + 4 * hi;
#pragma unroll
        for (int rp = 0; rp < RPW / 2; ++rp) {
            const int y = y_p[rp], img = img_p[rp];
            if (y >= a.H) continue;                                 // wave-uniform
            float4 pv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float u[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) u[j] = fmax_lane_xor1(fmax_raw(acc[2 * rp][tt][4 * q + j], acc[2 * rp + 1][tt][4 * q + j]));
                pv[q] = make_float4(u[0], u[1], u[2], u[3]);
            }
            if (x < a.W && !(x & 1)) {
                float* dst = static_cast<float*>(a.pool_out) + ((size_t)(img * Hp + (y >> 1)) * Wp + (x >> 1)) * a.Nout + nbase;
#pragma unroll
                for (int q = 0; q < 4; ++q) *reinterpret_cast<float4*>(dst + 8 * q) = pv[q];
            }
            if constexpr (PCODES) if (a.pool_codes_out != nullptr) {      // (wave-uniform) argmax of every window, ConvArgs::pool_codes_out; the 32- / 64-channel tiles of full-size problems only, like the slope codes
                // The winner is the first element EQUAL to the window maximum in the order top-left, top-right, bottom-left, bottom-right (unet_misc.hip
                // POOL_BWD_1).  Et / Eb: this lane's top / bottom value equals the maximum (one bit per element, built as E = 2 E + (v == max) from element
                // 15 down: v_cmp + v_addc, two instructions per bit and two live registers -- the first form of this epilogue compared the four values with
                // each other, three masks and ~13 instructions per element, and cost conv_x3w_kernel four spilled registers).
                unsigned Et = 0, Eb = 0;
#pragma unroll
                for (int i = 15; i >= 0; --i) {
                    const float mx = i % 4 == 0 ? pv[i / 4].x : (i % 4 == 1 ? pv[i / 4].y : (i % 4 == 2 ? pv[i / 4].z : pv[i / 4].w));
                    asm("v_cmp_eq_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(Et) : "v"(acc[2 * rp][tt][i]), "v"(mx) : "vcc");
                    asm("v_cmp_eq_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(Eb) : "v"(acc[2 * rp + 1][tt][i]), "v"(mx) : "vcc");
                }
                const unsigned Etr = (unsigned)__builtin_amdgcn_update_dpp(0, (int)Et, 0xB1, 0xF, 0xF, true);      // lane ^ 1: the right column's top (even lanes = left column)
                const unsigned Vb = ~(Et | Etr), Hb = ~Et & (Etr | ~Eb);                                            // bottom row / right column
                if (x < a.W && !(x & 1))
                    a.pool_codes_out[((((size_t)(img * Hp + (y >> 1)) * Wp + (x >> 1)) * (size_t)(a.Nout >> 5)) + (size_t)((nb * BN + tt * 32) >> 5)) * 2 + hi] = (Vb & 0xFFFFu) | (Hb << 16);
            }
        }
    }
}

typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void bdma16(i32x4 rsrc, unsigned voff, unsigned soff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(soff), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// Streamed fragment blocks (round 6).  A stage's MFMA work is a sequence of blocks b = (tap kx, pixel row r, channel block tt), six piece products on ONE
// accumulator each (smallest first: the per-accumulator summation order of the round-5 loops, so results are the same bits).  hipcc's own schedule of the
// round-5 loops -- with the register file full -- reused one fragment register set serially at the head of every tap (ds_read, s_waitcnt lgkmcnt(0), MFMA,
// four to five times per tap: an LDS round trip per MFMA) and left the matrix pipe idle for about as long as it ran.  Here the operands a block needs
// beyond its predecessor's (the pixel fragments when (kx, r) changes, the weight fragments when (kx, tt) changes) are read into the OTHER half of two
// small double buffers while the predecessor's six MFMAs run, and __builtin_amdgcn_sched_barrier pins that order: 2 x (12 + 12) fragment registers
// instead of 60 ... 72, no LDS latency in front of any MFMA but a stage's first.
//   readX(kx, r, X[3]) / readW(kx, tt, W[3]): the three piece fragments of pixel row r / channel block tt at tap kx.
// WREUSE (NT > 1 only): the blocks of a tap run (channel block tt, pixel row r) with r fastest, every pixel row's fragments resident (one buffer per row)
// and the weight fragments of a channel block used for all RPW rows before the next block's replace them: a stage then reads every pixel fragment once
// per tap (as before) and every weight fragment once per tap instead of RPW times -- 3 (RPW + NT) fragment triples instead of 3 RPW (1 + NT): -33 % of a
// 64-channel tile's LDS reads, -40 % of a 128-channel tile's, in the same 2 x (12 + 12) fragment registers at RPW = 2.  An accumulator still receives its
// taps in the order 0, 1, 2 and its six products smallest first: the same bits.
template <int RPW, int NT, bool WREUSE = false, typename RX, typename RW>
__device__ __forceinline__ void x3_stage_blocks(f32x16 (&acc)[RPW][NT], RX&& readX, RW&& readW) {
    constexpr int NBLK = 3 * RPW * NT;
    constexpr int WI[6] = {0, 1, 2, 0, 1, 0};
    constexpr int XI[6] = {2, 1, 0, 1, 0, 0};
    if constexpr (WREUSE && NT > 1) {
        uint4 X[RPW][3], Wf[2][3];
        readX(0, 0, X[0]);
        readW(0, 0, Wf[0]);
#pragma unroll
        for (int b = 0; b < NBLK; ++b) {
            const int tt = (b / RPW) % NT, r = b % RPW;
            const int wi = (b / RPW) & 1;
            if (b + 1 < NBLK) {
                const int kx1 = (b + 1) / (RPW * NT), tt1 = ((b + 1) / RPW) % NT, r1 = (b + 1) % RPW;
                if (tt1 == 0) readX(kx1, r1, X[r1]);                   // (row r1's buffer: its last reader, block (kx1 - 1, NT - 1, r1), has been issued)
                if (r1 == 0) readW(kx1, tt1, Wf[wi ^ 1]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 6; ++q)
                acc[r][tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, Wf[wi][WI[q]]), __builtin_bit_cast(bf16x8, X[r][XI[q]]), acc[r][tt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        return;
    }
    uint4 X[2][3], Wf[2][3];
    readX(0, 0, X[0]);
    readW(0, 0, Wf[0]);
#pragma unroll
    for (int b = 0; b < NBLK; ++b) {
        const int kx = b / (RPW * NT), r = (b / NT) % RPW, tt = b % NT;
        const int xi = (b / NT) & 1;                                   // the pixel fragments change every NT blocks,
        const int wi = NT > 1 ? (b & 1) : ((b / RPW) & 1);            // the weight fragments every block (NT > 1) or every tap (NT == 1)
        if (b + 1 < NBLK) {
            const int kx1 = (b + 1) / (RPW * NT), r1 = ((b + 1) / NT) % RPW, tt1 = (b + 1) % NT;
            if (kx1 != kx || r1 != r) readX(kx1, r1, X[xi ^ 1]);
            if (kx1 != kx || tt1 != tt) readW(kx1, tt1, Wf[wi ^ 1]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 6; ++q)
            acc[r][tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, Wf[wi][WI[q]]), __builtin_bit_cast(bf16x8, X[xi][XI[q]]), acc[r][tt], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
}

}  // namespace
