// conv_bfg.hip -- the bf16 2x2 / stride-2 transposed convolutions of the U-Net decoder (upv6..upv9, models/arch/Unet.py:30-42,68-83) as pixel GEMMs
// with BOTH operands staged by LDS-DMA, gfx950:
//   forward        out[2y+ty][2x+tx][co] = b[co] + sum_ci in[y][x][ci] W[ci][co][ty][tx]       GEMM  M = input pixels, K = Cin,    N = 4 Cout
//   backward-data  din[y][x][ci] = slope(act[y][x][ci]) sum_{t,co} dout[2y+ty][2x+tx][co] W[ci][co][t]   M = input pixels, K = 4 Cout, N = Cin
// Same contract and arithmetic as conv_igemm_kernel<bf16_t, CONV_1X1 / CONV_GATHER2X2> (conv_igemm.hip): bf16 NHWC tensors, fp32 accumulation on
// v_mfma_f32_32x32x16_bf16 in the same k order (32-channel chunks; backward: the four taps inside each chunk) -- the parity test demands identical bits.
//
// These launches are bandwidth problems (K <= 512 and the larger tensor is twice the smaller one: 85-340 flop per byte in bf16):
// the register-staged kernel spent its time in two barriers per 8 MFMAs and 8-byte scattered stores (2.1 TB/s).  Here:
//   * work item = (16 x 32 pixel tile, BN-wide column block, 32-channel k chunk): one 32 KB activation tile + one BN x 64 B weight slab, both
//     copied HBM/L2 -> LDS by linear 1 KiB LDS-DMA pieces in conv_bfd's XOR-swizzled landing order (conflict-free ds_read_b128 fragments);
//     the weights are packed once per step as the exact LDS image of every slab (bfg_store, unet_misc.hip);
//   * a ring of three (tile, slab) buffers: items j+1 and j+2 are in flight while item j is multiplied; the ring runs across column blocks and
//     tiles, so the queue never drains inside a launch; one barrier per item; waits are counted (every item has exactly A_IT + B_IT DMA
//     instructions per wave -- past the end of the work, out-of-range dummies -- so the count is one immediate);
//   * bias lives in LDS; outputs leave as full 64-byte runs per pixel (conv.h bf16_line_swap), forward scattered to the four taps' pixels.
#include <stdlib.h>
#include "conv.h"

#define TW 32
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

namespace {

__device__ __forceinline__ void bfg_dma16(i32x4 rsrc, unsigned voff, unsigned soff, unsigned lds_dst) {      // as conv_bfd.hip::bfd_dma16
    unsigned keep;
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(soff), "s"(lds_dst) : "memory");
}

constexpr int BFG_NAB = 3, BFG_WAVES = 8, BFG_RPW = 2, BFG_TH = BFG_WAVES * BFG_RPW;
constexpr int BFG_A_BYTES = BFG_TH * TW * 64;            // 32 KB: 512 pixels x one 32-channel chunk
constexpr int BFG_A_IT = BFG_A_BYTES / 1024 / BFG_WAVES; // 4 pieces per wave
constexpr int BFG_BIAS_BYTES = 2048;                     // up to 512 floats

// GATHER: false = forward (1x1 on the input image, scatter epilogue), true = backward-data (k runs over the 4 taps x Cout of the 2H x 2W gradient)
template <int BN, bool GATHER>
__global__ __launch_bounds__(64 * BFG_WAVES) void conv_bfg_kernel(const ConvArgs a) {
    constexpr int WAVES = BFG_WAVES, RPW = BFG_RPW, TH = BFG_TH, NT = BN / 32, A_BYTES = BFG_A_BYTES, A_IT = BFG_A_IT;
    constexpr int B_BYTES = BN * 64, B_PIECES = B_BYTES / 1024;
    constexpr int NAB = BFG_NAB;
    extern __shared__ __attribute__((aligned(16))) char lds[];           // [A0][A1][A2][B0][B1][B2][bias]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 31, hi = lane >> 5;
    const int Cs = a.C0;                                                  // channels of the A source tensor (Cin forward, Cout backward)
    const int NB = a.Nout / BN;
    const int NCHs = Cs >> 5;                                             // 32-channel chunks of the source tensor
    const int KC = GATHER ? 4 * NCHs : NCHs;                              // k chunks per (tile, column block)
    const int lKC = __builtin_ctz(KC), lNB = __builtin_ctz(NB);      // all powers of two (checked at launch)
    const int tiles_img = a.tiles_x * a.tiles_y;
    const int total_tiles = tiles_img * a.N;
    const int Hs = GATHER ? 2 * a.H : a.H, Ws = GATHER ? 2 * a.W : a.W;  // source image dims

    const unsigned lds_base = (unsigned)__builtin_amdgcn_readfirstlane((int)(size_t)(__attribute__((address_space(3))) char*)lds);
    const unsigned ldsB_addr = lds_base + NAB * A_BYTES;
    float* lds_bias = reinterpret_cast<float*>(lds + NAB * A_BYTES + NAB * B_BYTES);
    const unsigned long long wbase = (unsigned long long)a.wp;
    const i32x4 rsrc_w = {(int)(unsigned)wbase, (int)((unsigned)(wbase >> 32) & 0xFFFFu), (int)((size_t)KC * NB * B_BYTES), 0x00020000};
    constexpr unsigned OOB = 0xFFFFFFF0u;

    const int first = xcd_block(a.xcd), stride = gridDim.x;
    if (first >= total_tiles) return;
    if (!GATHER) {                               // bias -> LDS once (a global load in the loop's epilogue would drain the DMA queue: vmcnt retires in order)
        for (int i = tid; i < a.Cout_t; i += 64 * WAVES) lds_bias[i] = a.bias[i];
    }
    __syncthreads();

    // ---- activation DMA: this lane's unit of piece wave + it*WAVES: pixel P = u >> 2 of the tile, landing slot u & 3 = octet ^ ((P >> 2) & 3) ----
    int a_row[A_IT], a_col[A_IT];
    unsigned a_oct[A_IT];
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
        const int u = (wave + it * WAVES) * 64 + lane;
        const int P = u >> 2;
        a_row[it] = P >> 5; a_col[it] = P & 31;
        a_oct[it] = (unsigned)((u & 3) ^ ((P >> 2) & 3)) * 16u;
    }
    unsigned a_pix[A_IT];                        // source pixel index (tap (0,0) for the gather) of the unit in tile l_tile, OOB if outside the image
    int l_tile = -1, l_img = 0;
    auto decode = [&](int t, int& img, int& y0, int& x0) {
        img = t / tiles_img;
        const int r = t - img * tiles_img;
        const int ty = r / a.tiles_x, tx = r - ty * a.tiles_x;
        y0 = ty * TH; x0 = tx * TW;
    };
    auto setup_load = [&](int t) {
        int y0, x0;
        decode(t, l_img, y0, x0);
        l_tile = t;
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            const int gy = y0 + a_row[it], gx = x0 + a_col[it];
            const bool ok = gy < a.H && gx < a.W;
            a_pix[it] = ok ? (GATHER ? (unsigned)(4 * gy * a.W + 2 * gx) : (unsigned)(gy * a.W + gx)) : OOB;
        }
    };
    const int my_tiles = (total_tiles - first + stride - 1) / stride;
    const int n_items = my_tiles << (lNB + lKC);
    // item j = (tile first + (j >> (lNB + lKC)) * stride, column block (j >> lKC) & (NB - 1), k chunk j & (KC - 1)); ring slot j % 3.
    // Past the last item the same instructions are issued with out-of-range offsets (zeros into a slot nobody reads again): uniform counts.
    auto issue = [&](int j) {
        const int buf = j % NAB;
        const int kc = j & (KC - 1), nb = (j >> lKC) & (NB - 1);
        const int t = first + (j >> (lNB + lKC)) * stride;
        const bool live = j < n_items;
        if (live && t != l_tile) setup_load(t);
        const int tap = GATHER ? (kc & 3) : 0, chunk = GATHER ? (kc >> 2) : kc;      // backward: chunk-major, taps inside (conv_igemm's order)
        const int kslab = GATHER ? tap * NCHs + chunk : kc;                          // k chunk of the packed weights: k = tap * Cout + co
        const unsigned tapoff = GATHER ? (unsigned)((tap >> 1) * Ws + (tap & 1)) : 0u;
        const size_t img_bytes = (size_t)Hs * Ws * Cs * 2;
        const unsigned long long ab = (unsigned long long)(static_cast<const char*>(a.in0) + (size_t)l_img * img_bytes);
        const i32x4 rsrc_a = {(int)(unsigned)ab, (int)((unsigned)(ab >> 32) & 0xFFFFu), (int)img_bytes, 0x00020000};
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            const unsigned voff = (live && a_pix[it] != OOB) ? (a_pix[it] + tapoff) * (unsigned)(Cs * 2) + a_oct[it] : OOB;
            bfg_dma16(rsrc_a, voff, (unsigned)(chunk * 64), lds_base + (unsigned)(buf * A_BYTES + (wave + it * WAVES) * 1024));
        }
        {
            const int piece = wave % B_PIECES;
            const unsigned soff = live ? (unsigned)(((kslab << lNB) + nb) * B_BYTES + piece * 1024) : 0u;
            bfg_dma16(rsrc_w, live ? (unsigned)lane * 16u : OOB, soff, ldsB_addr + (unsigned)(buf * B_BYTES + piece * 1024));
        }
    };

    // ---- fragment addresses ------------------------------------------------------------------------------------------------------------
    unsigned f_off[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) f_off[kb] = (unsigned)(m * 64 + (((kb * 2 + hi) ^ ((m >> 2) & 3)) * 16));

    issue(0);
    issue(1);
    f32x16 acc[RPW][NT];
    for (int j = 0; j < n_items; ++j) {
        const int kc = j & (KC - 1), nb = (j >> lKC) & (NB - 1);
        const int t = first + (j >> (lNB + lKC)) * stride;
        eld_wait_vmcnt<A_IT + 1>();              // this wave's pieces of item j have landed (item j+1's A_IT + 1 may still fly; see conv_bfs.hip on the count)
        __syncthreads();                         // ... and everybody else's; everybody is done reading ring slot (j + 2) % 3
        issue(j + 2);
        if (kc == 0) {
#pragma unroll
            for (int r = 0; r < RPW; ++r)
#pragma unroll
                for (int tt = 0; tt < NT; ++tt)
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[r][tt][i] = 0.f;
        }
        const char* la = lds + (j % NAB) * A_BYTES + (wave * RPW) * (TW * 64);
        const char* lb = lds + NAB * A_BYTES + (j % NAB) * B_BYTES;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            uint4 fx[RPW], fw[NT];
#pragma unroll
            for (int r = 0; r < RPW; ++r) fx[r] = *reinterpret_cast<const uint4*>(la + r * (TW * 64) + f_off[kb]);
#pragma unroll
            for (int tt = 0; tt < NT; ++tt) fw[tt] = *reinterpret_cast<const uint4*>(lb + tt * (32 * 64) + f_off[kb]);
#pragma unroll
            for (int r = 0; r < RPW; ++r)
#pragma unroll
                for (int tt = 0; tt < NT; ++tt)
                    acc[r][tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fw[tt]), __builtin_bit_cast(bf16x8, fx[r]), acc[r][tt], 0, 0, 0);      // D[channel][pixel]
        }
        if (kc + 1 < KC) continue;

        // ---- epilogue of (tile t, column block nb): full-line layout (conv.h bf16_line_swap): lane l holds pixel (l & 15) + 16 i, group bf16_line_group(l) ----
        int img, y0, x0;
        decode(t, img, y0, x0);
        const int lp = lane & 15, lg = bf16_line_group(lane);
        const bool second = x0 + 16 < a.W;
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const int y = y0 + wave * RPW + r;                          // wave-uniform
            const bool yok = y < a.H;
            const int yc = yok ? y : a.H - 1;
#pragma unroll
            for (int tt = 0; tt < NT; ++tt) {
                const int n0 = nb * BN + tt * 32;
                float4 v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = make_float4(acc[r][tt][4 * q], acc[r][tt][4 * q + 1], acc[r][tt][4 * q + 2], acc[r][tt][4 * q + 3]);
                bf16_t* d0;                                              // destination of pixel x0 + lp (i = 0); pixel + 16 is dstep elements further
                size_t dstep;
                if (!GATHER) {                                           // EPI_CONVT_FWD: n = tap * Cout + co; a 32-block never straddles a tap
                    const int lC = __builtin_ctz(a.Cout_t);
                    const int tap = n0 >> lC, co0 = n0 & (a.Cout_t - 1);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 bq = *reinterpret_cast<const float4*>(lds_bias + co0 + 4 * hi + 8 * q);
                        v[q].x += bq.x; v[q].y += bq.y; v[q].z += bq.z; v[q].w += bq.w;
                    }
                    d0 = static_cast<bf16_t*>(a.out0) + ((size_t)(img * 2 * a.H + 2 * yc + (tap >> 1)) * (2 * a.W) + 2 * (x0 + lp) + (tap & 1)) * a.Cout_t + co0 + 8 * lg;
                    dstep = (size_t)32 * a.Cout_t;
                } else {                                                 // EPI_GRAD into one tensor of Nout channels, times the slope of the saved activation
                    const size_t e0 = ((size_t)(img * a.H + yc) * a.W + x0) * a.Nout + n0 + 8 * lg;
                    d0 = static_cast<bf16_t*>(a.out0) + e0 + (size_t)lp * a.Nout;
                    dstep = (size_t)16 * a.Nout;
                    if (a.act0 != nullptr) {
                        const bf16_t* arow = static_cast<const bf16_t*>(a.act0) + e0;
                        const uint4 a0 = *reinterpret_cast<const uint4*>(arow + (size_t)min(lp, a.W - 1 - x0) * a.Nout);
                        const uint4 a1 = *reinterpret_cast<const uint4*>(arow + (size_t)min(lp + 16, a.W - 1 - x0) * a.Nout);
                        uint2 sp[4];
                        bf16_line_unswap(a0, a1, sp);
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float4 sv = unpack_bf4(sp[q]);
                            v[q].x *= lrelu_slope(sv.x); v[q].y *= lrelu_slope(sv.y); v[q].z *= lrelu_slope(sv.z); v[q].w *= lrelu_slope(sv.w);
                        }
                    }
                }
                uint2 pk[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) pk[q] = pack_bf4(v[q]);
                uint4 s0, s1;
                bf16_line_swap(pk, s0, s1);                              // every lane takes part; only the stores are predicated
                if (yok) {
                    if (x0 + lp < a.W) *reinterpret_cast<uint4*>(d0) = s0;
                    if (second && x0 + lp + 16 < a.W) *reinterpret_cast<uint4*>(d0 + dstep) = s1;
                }
            }
        }
    }
}

template <int BN, bool GATHER>
int launch_bfg(ConvArgs a, hipStream_t st) {
    a.xcd = eld_xcd_mask() & XCD_BF16;
    a.tiles_x = (a.W + TW - 1) / TW;
    a.tiles_y = (a.H + BFG_TH - 1) / BFG_TH;
    const size_t lds_bytes = (size_t)BFG_NAB * (BFG_A_BYTES + BN * 64) + BFG_BIAS_BYTES;
    const long long tiles = (long long)a.tiles_x * a.tiles_y * a.N;
    if (tiles <= 0) return 0;
    if (tiles > 0x00ffffffLL) return ELD_ENOTSUP;
    auto kern = conv_bfg_kernel<BN, GATHER>;
    static EldAttrOnce once;
    { const int rc = once.ensure(kern, lds_bytes); if (rc) return rc; }
    long long grid = (long long)eld_num_cus();
    if (grid > tiles) grid = tiles;
    ELD_LAUNCH(kern, dim3((unsigned)grid), dim3(64 * BFG_WAVES), lds_bytes, st, a);
    ELD_LAUNCH_CHECK();
    return 0;
}

inline bool pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

}  // namespace

// Column-block width the transposed-conv launch (GEMM N = Nout, source tensor of Cs channels on an N x H x W pixel domain) takes on conv_bfg_kernel,
// 0 if it stays on conv_igemm_kernel<bf16_t>: powers of two only (the item decoding shifts), every CU a tile.  gather: backward-data.
int bfg_slab_bn(bool gather, int Nout, int Cs, int Cout_t, int N, int H, int W) {
    if (debug_kernel_mask(-1) & 2) return 0;
    if (!pow2(Nout) || !pow2(Cs) || Cs < 32 || Nout < 64 || Cs > 512) return 0;
    if (!gather && (!pow2(Cout_t) || Cout_t < 32 || Cout_t > 512 || Nout != 4 * Cout_t || Nout < 128)) return 0;
    const long long px_tiles = (long long)((W + TW - 1) / TW) * ((H + BFG_TH - 1) / BFG_TH) * N;
    if (2 * px_tiles < eld_num_cus()) return 0;      // (the workgroup loops over the column blocks of its tile: half the CUs busy still beats the generic kernel)
    const size_t src_bytes = (size_t)(gather ? 4 : 1) * H * W * Cs * 2;
    if (src_bytes >= 0xFFFFFFF0ull) return 0;
    return Nout % 128 == 0 ? 128 : 64;
}

int launch_conv_bfg(const ConvArgs& a, int mode, hipStream_t st) {
    const bool gather = mode == CONV_GATHER2X2;
    const int bn = bfg_slab_bn(gather, a.Nout, a.C0, a.Cout_t, a.N, a.H, a.W);
    if (!bn || a.C1 != 0) return ELD_ENOTSUP;
    if (!gather) {
        if (a.epi != EPI_CONVT_FWD || bn != 128) return ELD_ENOTSUP;
        return launch_bfg<128, false>(a, st);
    }
    if (a.epi != EPI_GRAD || a.split != a.Nout || a.out1 != nullptr) return ELD_ENOTSUP;
    return bn == 128 ? launch_bfg<128, true>(a, st) : launch_bfg<64, true>(a, st);
}
