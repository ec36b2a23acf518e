// conv_bfw.hip -- bf16 3x3 convolution for the 64-OUTPUT-CHANNEL layers of the U-Net with K = 32 / 64 input channels (half resolution: conv2_1,
// conv2_2, conv8_2 forward; the backward-data of conv2_2, conv8_2 and conv9_1), gfx950.  Same contract as conv_bfd_kernel (conv_bfd.hip): bf16 NHWC
// activations, weights in conv_bfd's slab layout at BN = 64, fp32 accumulation on v_mfma_f32_32x32x16_bf16, bias / LeakyReLU / slope / fused 2x2
// max-pool epilogues -- models/arch/Unet.py:13-16,37-40,52-53,91-98 and their autograd backward-data.
//
// Why a third kernel: on conv_bfd_kernel<64> these layers stream a 12 KB weight slab per (chunk, kernel row) stage of only 24 MFMAs per wave
// (0.8 us -- the slab ring's two-stage lead is shorter than a DMA round trip) and move 75 KB per 72 MFMAs through a per-CU memory path that
// sustains ~25 GB/s: 34 % matrix-pipe busy (profiles/r03_pmc_bf16.md).  With K <= 64 the layer's WHOLE weight tensor is 72 KB:
//   * weights (9 taps x 64 x K, the pack kernel's slab image) are copied into LDS once per workgroup and stay there; the main loop's only
//     LDS-DMA traffic is the activation halo tile;
//   * a work item is (tile, 16-channel HALF chunk): halo tiles of 18 x 34 pixels x 16 channels = 20 KB run through a ring of FOUR buffers, so
//     the tiles of items j+1 .. j+3 are in flight while item j is multiplied (a 3.7 us lead at 36 MFMAs per wave and item).  A pixel of a
//     half chunk is 32 bytes (two 16-byte units, their order XOR-swizzled by the halo column so that the lane groups of a ds_read_b128 fall on
//     distinct banks for every tap shift);
//   * one barrier per item; every wave waits for its own pieces with the counted, conservative s_waitcnt of conv_bfs.hip.
// The K order of the accumulation is (32-channel chunk, 16-channel half, ky, kx) instead of conv_igemm / conv_bfd's (chunk, ky, kx, half): results
// equal theirs up to the fp32 summation order (tests compare with a bf16-rounding tolerance; the other DMA kernels are bit-identical to conv_igemm).
#include <stdlib.h>
#include "conv.h"

#define TW 32
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {

__device__ __forceinline__ void bfw_dma16(i32x4 rsrc, unsigned voff, unsigned soff, unsigned lds_dst) {      // as conv_bfd.hip::bfd_dma16
    unsigned keep;
    soff = (unsigned)__builtin_amdgcn_readfirstlane((int)soff);        // wave-uniform by construction; the "s" constraint alone does not force an SGPR
    lds_dst = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_dst);
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(soff), "s"(lds_dst) : "memory");
}

constexpr int BFW_NAB = 4;                               // activation ring depth
constexpr int BFW_WCHUNK = 9 * 64 * 64;                  // packed weights of one 32-channel chunk (BN = 64): 36 DMA pieces

// ACT: EPI_GRAD with at least one saved activation (slope epilogue) -- compile time, because the activation loads are hand-issued asm loads whose
// destination registers must not pass through a phi (see conv_bfs.hip)
template <bool ACT>
__global__ __launch_bounds__(512) void conv_bfw_kernel(const ConvArgs a) {
    constexpr int RPW = 2, WAVES = 8, NT = 2;
    constexpr int TH = WAVES * RPW, HW2 = TW + 2, A_PIX = (TH + 2) * HW2;
    constexpr int A_UNITS = A_PIX * 2, A_PIECES = (A_UNITS + 63) / 64, A_BYTES = A_PIECES * 1024;      // 1224 units -> 20 pieces
    constexpr int A_IT = (A_PIECES + WAVES - 1) / WAVES;                                                 // 3
    constexpr int ROWB = HW2 * 32;                                                                       // bytes of a halo row
    extern __shared__ __attribute__((aligned(16))) char lds[];           // [A0][A1][A2][A3][W chunk 0][W chunk 1]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 31, hi = lane >> 5;
    const int Cin = a.C0 + a.C1, NCH = Cin >> 5, NCH0 = a.C0 >> 5;       // NCH = 1 or 2
    const int NI = 2 * NCH;                                               // items per tile
    const int tiles_img = a.tiles_x * a.tiles_y;
    const int total_tiles = tiles_img * a.N;
    const int Cs0 = a.C0;

    const unsigned lds_base = (unsigned)__builtin_amdgcn_readfirstlane((int)(size_t)(__attribute__((address_space(3))) char*)lds);
    const unsigned ldsW_addr = lds_base + BFW_NAB * A_BYTES;
    constexpr unsigned OOB = 0xFFFFFFF0u;

    // ---- the layer's weights: 36 * NCH linear 1 KiB pieces, dealt round-robin (a duplicate piece rewrites identical bytes) -------------------
    {
        const unsigned long long wbase = (unsigned long long)a.wp;
        const i32x4 rsrc_w = {(int)(unsigned)wbase, (int)((unsigned)(wbase >> 32) & 0xFFFFu), NCH * BFW_WCHUNK, 0x00020000};
        const int wpieces = 36 * NCH;
#pragma unroll
        for (int it = 0; it < 9; ++it) {                                  // 72 / 8 = 9; with NCH = 1 the later ones repeat earlier pieces
            const int piece = (wave + it * WAVES) % wpieces;
            bfw_dma16(rsrc_w, (unsigned)lane * 16u, (unsigned)(piece * 1024), ldsW_addr + (unsigned)(piece * 1024));
        }
    }

    // ---- activation DMA: which (halo pixel, octet of the half chunk) lands in this lane's slot of piece wave + it*WAVES --------------------
    // The landing order is linear (lane l of a piece at 16 l), so the LDS layout is chosen by WHICH unit a lane fetches: pixel P = u / 2 of the
    // halo tile, octet (u & 1) ^ f of its 16 channels, f = (halo column >> 3) & 1.  A ds_read_b128 is serviced in lane groups {0-3,12-15,20-27},
    // {4-11,16-19,28-31} (+32): their columns pair up at distances 8 and 24, where f differs, so the 16 lanes of a group fall on 16 distinct
    // 16-byte slots of the 256-byte bank row for every tap shift (without f: two lanes per slot, every activation read 2-way conflicted).
    int a_hy[A_IT], a_hx[A_IT];
    unsigned a_oct[A_IT];
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
        const int piece = (wave + it * WAVES) % A_PIECES;
        const int u = piece * 64 + lane;
        const int P = u >> 1;
        const int hr = P / HW2, hc = P - hr * HW2;
        a_hy[it] = u < A_UNITS ? hr - 1 : -1000;
        a_hx[it] = hc - 1;
        a_oct[it] = (unsigned)((u & 1) ^ ((hc >> 3) & 1)) * 16u;
    }
    unsigned a_voff[A_IT];
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
            const int gy = y0 + a_hy[it], gx = x0 + a_hx[it];
            const bool ok = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
            a_voff[it] = ok ? (unsigned)(gy * a.W + gx) * (unsigned)(Cs0 * 2) + a_oct[it] : OOB;
        }
    };
    const int first = xcd_block(a.xcd), stride = gridDim.x;
    if (first >= total_tiles) return;
    const int my_tiles = (total_tiles - first + stride - 1) / stride;
    const int n_items = my_tiles * NI;
    // item j = (tile first + (j / NI) * stride, half chunk j % NI); its halo tile lives in ring slot j % 4.  Past the last item the same A_IT
    // instructions are issued with out-of-range offsets (zeros land in a slot nobody reads again): every wait of the loop is the same immediate.
    auto issue_A = [&](int j) {
        const int k = NCH == 2 ? (j >> 2) : (j >> 1), sub = NCH == 2 ? (j & 3) : (j & 1);
        const int chunk = sub >> 1, half = sub & 1;
        const int t = first + k * stride;
        if (j >= n_items) {
#pragma unroll
            for (int it = 0; it < A_IT; ++it) a_voff[it] = OOB;
            l_tile = -1;
        } else if (t != l_tile) setup_load(t);
        const char* src = static_cast<const char*>(chunk < NCH0 ? a.in0 : a.in1);
        const int cs = chunk < NCH0 ? chunk : chunk - NCH0;
        const size_t img_bytes = (size_t)a.H * a.W * Cs0 * 2;
        const unsigned long long ab = (unsigned long long)(src + (size_t)l_img * img_bytes);
        const i32x4 rsrc_a = {(int)(unsigned)ab, (int)((unsigned)(ab >> 32) & 0xFFFFu), (int)img_bytes, 0x00020000};
        const int buf = j & (BFW_NAB - 1);
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            const int piece = (wave + it * WAVES) % A_PIECES;
            bfw_dma16(rsrc_a, a_voff[it], (unsigned)(cs * 64 + half * 32), lds_base + (unsigned)(buf * A_BYTES + piece * 1024));
        }
    };

    // ---- fragment addresses -----------------------------------------------------------------------------------------------------------------
    unsigned fx_off[3];                                                   // + row * ROWB
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) fx_off[kx] = (unsigned)((m + kx) * 32 + ((hi ^ (((m + kx) >> 3) & 1)) * 16));
    unsigned fw_off[2];                                                   // slab rows of 64 B, units XOR-swizzled (conv_bfd.hip): half = k-block
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) fw_off[kb] = (unsigned)(m * 64 + (((kb * 2 + hi) ^ ((m >> 2) & 3)) * 16));
    const char* ldsW = lds + BFW_NAB * A_BYTES;

    // bias -> LDS once per workgroup (a global load inside the loop would be waited for with a vmcnt that drains the DMA queue ahead of it).  The
    // accumulators of a tile START from it, so the epilogue has no bias add.
    float* lds_bias = reinterpret_cast<float*>(lds + BFW_NAB * A_BYTES + 2 * BFW_WCHUNK);
    if constexpr (!ACT) {
        if (tid < 64) lds_bias[tid] = a.epi == EPI_FWD ? a.bias[tid] : 0.f;      // (published by the first barrier of the loop)
    }
    const float sl = a.lrelu ? 0.2f : 1.0f;                // max(1 v, v) = v
    // destination / saved-activation tensors of the two 32-channel blocks (EPI_GRAD may split them over two tensors: conv9_1's backward)
    const bf16_t* actp[NT]; bf16_t* outp[NT]; int Cd[NT], cbd[NT];
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) {
        if (a.epi == EPI_FWD) { outp[tt] = static_cast<bf16_t*>(a.out0); Cd[tt] = 64; cbd[tt] = tt * 32; actp[tt] = nullptr; }
        else {
            const bool lo = tt * 32 < a.split;
            outp[tt] = static_cast<bf16_t*>(lo ? a.out0 : a.out1);
            Cd[tt] = lo ? a.split : 64 - a.split;
            cbd[tt] = lo ? tt * 32 : tt * 32 - a.split;
            actp[tt] = static_cast<const bf16_t*>(lo ? a.act0 : a.act1);
        }
    }

    issue_A(0);
    issue_A(1);
    issue_A(2);

    f32x16 acc[RPW][NT];
    // all nine taps of one 16-channel half chunk: 36 MFMAs per wave
    auto multiply = [&](int j, int chunk, int half, bool late) {
        if (ELD_DBG(a) & 2) { if (late) issue_A(j + 3); return; }
        const char* la0 = lds + (j & (BFW_NAB - 1)) * A_BYTES + (wave * RPW) * ROWB;
        const char* lw0 = ldsW + chunk * (BFW_WCHUNK / 3) + fw_off[half];      // slab(ky, chunk) at (ky * NCH + chunk) * 12288
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const char* la = la0 + ky * ROWB;
            const char* lw = lw0 + ky * NCH * (BFW_WCHUNK / 3);
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                uint4 fx[RPW], fw[NT];
#pragma unroll
                for (int r = 0; r < RPW; ++r) fx[r] = *reinterpret_cast<const uint4*>(la + r * ROWB + fx_off[kx]);
#pragma unroll
                for (int tt = 0; tt < NT; ++tt) fw[tt] = *reinterpret_cast<const uint4*>(lw + (kx * 64 + tt * 32) * 64);
#pragma unroll
                for (int r = 0; r < RPW; ++r)
#pragma unroll
                    for (int tt = 0; tt < NT; ++tt)
                        acc[r][tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fw[tt]), __builtin_bit_cast(bf16x8, fx[r]), acc[r][tt], 0, 0, 0);      // D[channel][pixel]
            }
            if (ky == 0 && late) issue_A(j + 3);
        }
    };
    // epilogue of tile t (layouts: conv_bfd.hip); ac = the saved activations of the tile's pixels in the line layout (ACT)
    auto epilogue = [&](int t, f32x16 (&R)[RPW][NT], u32x4 (&ac)[RPW][NT][2]) {
        if (ELD_DBG(a) & 1) return;
        int img, y0, x0;
        decode(t, img, y0, x0);
        const int lp = lane & 15, lg = bf16_line_group(lane);
        if (!ACT && a.epi == EPI_FWD) {          // max(0.2 v, v) once, in place (the pooled copy below reuses the activated values)
#pragma unroll
            for (int r = 0; r < RPW; ++r)
#pragma unroll
                for (int tt = 0; tt < NT; ++tt)
#pragma unroll
                    for (int q = 0; q < 4; ++q) lrelu4(R[r][tt], 4 * q, sl);
        }
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const int y = y0 + wave * RPW + r;
#pragma unroll
            for (int tt = 0; tt < NT; ++tt) {
                float4 v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = make_float4(R[r][tt][4 * q], R[r][tt][4 * q + 1], R[r][tt][4 * q + 2], R[r][tt][4 * q + 3]);
                if constexpr (ACT) {                                        // EPI_GRAD: times the LeakyReLU slope of the saved activation
                    if (actp[tt] != nullptr) {
                        uint2 sp[4];
                        bf16_line_unswap(make_uint4(ac[r][tt][0][0], ac[r][tt][0][1], ac[r][tt][0][2], ac[r][tt][0][3]),
                                         make_uint4(ac[r][tt][1][0], ac[r][tt][1][1], ac[r][tt][1][2], ac[r][tt][1][3]), sp);
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
                bf16_line_swap(pk, s0, s1);                                 // every lane takes part; only the stores are predicated
                bf16_t* row = outp[tt] + ((size_t)(img * a.H + y) * a.W + x0) * Cd[tt] + cbd[tt] + 8 * lg;
                if (y < a.H && x0 + lp < a.W) *reinterpret_cast<uint4*>(row + (size_t)lp * Cd[tt]) = s0;
                if (y < a.H && x0 + lp + 16 < a.W) *reinterpret_cast<uint4*>(row + (size_t)(lp + 16) * Cd[tt]) = s1;
            }
        }
        // fused nn.MaxPool2d(2) (Unet.py:53): vertical pair in the lane's own rows, horizontal pair in lane ^ 1, on the activated fp32 values (max
        // commutes with the monotone bf16 rounding, so this equals pooling the stored tensor)
        if (!ACT && a.epi == EPI_FWD && a.pool_out != nullptr) {
            const int x = x0 + m;
            const int Hp = a.H >> 1, Wp = a.W >> 1;
            const int y = y0 + wave * RPW;
            if (y < a.H) {
#pragma unroll
                for (int tt = 0; tt < NT; ++tt) {
                    uint2 pk[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float u[4];
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) u[jj] = fmax_lane_xor1(fmax_raw(R[0][tt][4 * q + jj], R[1][tt][4 * q + jj]));
                        pk[q] = pack_bf4(make_float4(u[0], u[1], u[2], u[3]));
                    }
                    const uint4 w0 = bf16_pair_swap(pk[0], pk[1]), w1 = bf16_pair_swap(pk[2], pk[3]);       // (every lane takes part)
                    if (x < a.W && !(x & 1)) {
                        bf16_t* dp = static_cast<bf16_t*>(a.pool_out) + ((size_t)(img * Hp + (y >> 1)) * Wp + (x >> 1)) * 64 + tt * 32 + 8 * hi;
                        *reinterpret_cast<uint4*>(dp) = w0;
                        *reinterpret_cast<uint4*>(dp + 16) = w1;
                    }
                }
            }
        }
    };

    for (int j = 0; j < n_items; ++j) {
        const int k = NCH == 2 ? (j >> 2) : (j >> 1), sub = NCH == 2 ? (j & 3) : (j & 1);
        const int chunk = sub >> 1, half = sub & 1;
        const int t = first + k * stride;
        const bool last_sub = sub + 1 == NI;
        // this wave's pieces of item j (and, the first time, of the weights) have landed; those of items j+1, j+2 may still fly
        // (the conservative count: it also waits for most of the previous epilogue's stores; counting them exactly and letting them fly measured the
        // same, profiles/r03_ab_notes.md)
        eld_wait_vmcnt<2 * A_IT>();
        __syncthreads();                         // ... and everybody else's; everybody is done reading ring slot (j - 1) % 4 = (j + 3) % 4
        const bool late = wave >= 4;             // waves 4-7 (the SIMD partners of 0-3) issue item j+3's pieces behind the first kernel row: see conv_bfd.hip
        if constexpr (ACT) {
            if (last_sub) {
                // The tile's last item, with the saved activations of its output pixels (line layout) loaded by hand so that the compiler does not
                // wait for them with a vmcnt that would drain the younger halo DMAs: they are OLDER than item j+3's pieces, so the wait below leaves
                // exactly those A_IT pieces in flight.  Out-of-range pixels re-read a valid address (the value is not used).  Load, multiply, wait
                // and epilogue sit in ONE branch and the destination registers live only here: no phi, no copy of a register that has not landed.
                u32x4 ac[RPW][NT][2];            // (tuples without a saved activation stay unwritten and unread)
                int img, y0, x0;
                decode(t, img, y0, x0);
                // A block without a saved activation loads a dummy line instead of skipping the statement: the count of hand-issued loads stays
                // constant, and no destination register is written on one path only (two paths writing the same variable would be merged through
                // copies of registers that have not landed -- the audit in tests/test_abi.py checks the generated assembly for exactly that).
#pragma unroll
                for (int r = 0; r < RPW; ++r) {
                    const int y = min(y0 + wave * RPW + r, a.H - 1);
                    const size_t rowp = (size_t)(img * a.H + y) * a.W;
#pragma unroll
                    for (int tt = 0; tt < NT; ++tt)
#pragma unroll
                        for (int i = 0; i < 2; ++i) {
                            const int xx = min(x0 + (lane & 15) + 16 * i, a.W - 1);
                            const bf16_t* p = actp[tt] != nullptr ? actp[tt] + (rowp + xx) * Cd[tt] + cbd[tt] + 8 * bf16_line_group(lane) : static_cast<const bf16_t*>(a.wp);
                            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(ac[r][tt][i]) : "v"(p) : "memory");
                        }
                }
                if (!late) issue_A(j + 3);
                multiply(j, chunk, half, late);
                // (whole 128-bit tuples as operands: see conv_bfs.hip)
                asm volatile("s_waitcnt vmcnt(%8)"
                             : "+v"(ac[0][0][0]), "+v"(ac[0][0][1]), "+v"(ac[0][1][0]), "+v"(ac[0][1][1]), "+v"(ac[1][0][0]), "+v"(ac[1][0][1]), "+v"(ac[1][1][0]), "+v"(ac[1][1][1])
                             : "n"(A_IT));
                epilogue(t, acc, ac);
                continue;
            }
        }
        if (!late) issue_A(j + 3);
        if (sub == 0) {                          // a tile's accumulators start from the bias (0 in the backward launches)
#pragma unroll
            for (int tt = 0; tt < NT; ++tt)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
                    if constexpr (!ACT) b = *reinterpret_cast<const float4*>(lds_bias + tt * 32 + 4 * hi + 8 * q);
#pragma unroll
                    for (int r = 0; r < RPW; ++r) { acc[r][tt][4 * q] = b.x; acc[r][tt][4 * q + 1] = b.y; acc[r][tt][4 * q + 2] = b.z; acc[r][tt][4 * q + 3] = b.w; }
                }
        }
        multiply(j, chunk, half, late);
        if constexpr (!ACT) {
            if (last_sub) {
                u32x4 none[RPW][NT][2];
                epilogue(t, acc, none);
            }
        }
    }
}

template <bool ACT>
int launch_bfw(ConvArgs a, hipStream_t st) {
    a.xcd = eld_xcd_mask() & XCD_BF16;
    constexpr int TH = 16;
    a.tiles_x = (a.W + TW - 1) / TW;
    a.tiles_y = (a.H + TH - 1) / TH;
    constexpr size_t A_BYTES = (size_t)(((TH + 2) * (TW + 2) * 2 + 63) / 64) * 1024;
    const size_t lds_bytes = BFW_NAB * A_BYTES + 2 * (size_t)BFW_WCHUNK + 256;      // + bias (the second weight chunk's space stays unused at K = 32)
    const long long tiles = (long long)a.tiles_x * a.tiles_y * a.N;
    if (tiles <= 0) return 0;
    if (tiles > 0x1fffffffLL) return ELD_ENOTSUP;
    auto kern = conv_bfw_kernel<ACT>;
    static EldAttrOnce once;
    { const int rc = once.ensure(kern, lds_bytes); if (rc) return rc; }
    long long grid = (long long)eld_num_cus();
    if (grid > tiles) grid = tiles;
    ELD_LAUNCH(kern, dim3((unsigned)grid), dim3(512), lds_bytes, st, a);
    ELD_LAUNCH_CHECK();
    return 0;
}

}  // namespace

// Layers this kernel takes from conv_bfd_kernel<64>: bf16 3x3 with exactly 64 output channels (one tensor, or EPI_GRAD split 32 + 32 over two) and
// K = 32 or 64 input channels, on a tile domain that gives every CU a tile.  Weights: conv_bfd's slab layout at BN = 64 (what bfd_slab_bn returns
// for these layers), so the choice between the two kernels is made per launch.
bool bfw_takes(const ConvArgs& a) {
    if (debug_kernel_mask(-1) & 8) return false;
    const int K = a.C0 + a.C1;
    if (a.Nout != 64 || (K != 32 && K != 64)) return false;
    if (a.C1 != 0 && a.C1 != a.C0) return false;
    if (a.epi == EPI_GRAD) { if (a.split != 64 && a.split != 32) return false; if (a.split == 32 && a.out1 == nullptr) return false; }
    else if (a.epi != EPI_FWD) return false;
    const long long px_tiles = (long long)((a.W + TW - 1) / TW) * ((a.H + 15) / 16) * a.N;
    return px_tiles >= eld_num_cus();
}

int launch_conv_bfw(const ConvArgs& a, hipStream_t st) {
    if ((size_t)a.H * a.W * a.C0 * 2 >= 0xFFFFFFF0ull) return ELD_ENOTSUP;
    if (a.pool_out && (a.epi != EPI_FWD || (a.H & 1) || (a.W & 1))) return ELD_EINVAL;
    if (a.epi == EPI_GRAD && (a.act0 != nullptr || a.act1 != nullptr)) return launch_bfw<true>(a, st);
    return launch_bfw<false>(a, st);
}
