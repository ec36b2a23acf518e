// conv_bfd.hip -- bf16 3x3 convolution (forward and backward-data) with BOTH operands staged by LDS-DMA, gfx950.
//
// Same contract as conv_igemm_kernel<bf16_t, CONV_3X3, ...> (conv_igemm.hip): bf16 NHWC activations, packed bf16 weights, fp32
// accumulation on v_mfma_f32_32x32x16_bf16, bias / LeakyReLU / slope epilogues in fp32, bf16 results -- BASELINE.json configs[2]
// ("bf16 U-Net with MFMA convs"), replacing nn.Conv2d(3x3, p=1) + max(0.2x, x) of models/arch/Unet.py:11-46,102-104 and its autograd
// backward-data.  What differs from the register-staged kernel is how operands reach LDS:
//
//   * bf16 operands need no conversion on the way in, so neither the activation halo tile nor the weight slab touches a VGPR or the
//     VALU: every 16-byte unit travels HBM/L2 -> LDS with buffer_load_dwordx4 ... lds (one wave instruction = 1 KiB, lane l lands at
//     LDS base + 16 l).  The landing order is linear, so the LDS layout is chosen by WHICH unit a lane fetches:
//         unit (pixel P, channel octet o of the 32-channel chunk)  ->  16-byte slot 4 P + (o ^ f),   f = (column >> 2) & 3
//     (column = halo column for activations, slab row for weights).  With that XOR the 16 lanes a ds_read_b128 services together
//     (lanes {0-3,12-15,20-27}, ...: MI355X_MICROARCH.md, LDS) fall on 16 distinct slots of the 256-byte bank row for every tap
//     shift: conflict-free fragment reads without padding.  Zero padding of the image border = out-of-range buffer offsets.
//   * weights are packed once per step as the exact LDS image of a stage's slab (bfd_store, unet_misc.hip):
//         slab(ky, chunk, nb) = [kx 0..2][n 0..BN)[4 units, swizzled]   = 3*BN*64 B at ((ky * K/32 + chunk) * Nout/BN + nb) * 3*BN*64
//   * stage = (32-channel chunk, kernel row ky): one barrier per stage.  A stage is only 48 MFMAs per wave (~1.5 us), shorter than a
//     DMA round trip under load (~2.7 us measured), so the weight slabs run TWO stages ahead in a ring of three buffers and the halo
//     tile of the next chunk / next tile three stages ahead (double buffer).  Each wave waits for its own pieces with a counted
//     s_waitcnt vmcnt(n), n = the DMA instructions it issued during the previous stage (every wave issues the same number: pieces are
//     dealt round-robin modulo the piece count, a duplicate piece rewrites identical bytes), then the barrier publishes them.
//     3 x 24 KB (BN = 128) + 2 x 39 KB (18 x 34 pixel halo, 64 B per pixel) = 150 KB, one 8-wave workgroup per CU.
//   * tile = 16 rows x 32 pixels x BN channels; a wave owns 2 rows x BN channels (2 x 4 accumulator tiles): each k-block reads
//     2 + 4 fragments for 8 MFMAs -> 36 KB of LDS reads per wave and stage against 48 x 32 matrix-pipe cycles (LDS 256 B/clk: 37 %).
//
// Round 4 -- no MFMA work on pixels that do not exist.  The workgroup still owns 512 pixel slots (8 waves x 2 MFMA columns-of-32), but which pixel
// a slot is follows from the LAUNCH's tile shape (a.tile_h rows x a.tile_w columns, tile_h * tile_w <= 512, (tile_h + 2)(tile_w + 2) <= 612 halo
// pixels): slot p is tile pixel (p / tile_w, p % tile_w), i.e. the M index -> (row, column) map is per-lane LDS addressing (12 fragment offsets
// instead of 6) and per-lane epilogue addressing.  Rows are rows of the virtual strip of the batch (conv.h vrow_*).  With 16 x 32 tiles cut per
// image, the 89 x 133 / 178 x 266 / 356 x 532 levels of a 1424 x 2128 frame executed 1.30x / 1.17x / 1.06x their pixels; with the shapes
// conv_tile_shape() picks (15 x 34, 17 x 30, 18 x 28 on the strip) it is 1.04x / 1.03x / 1.02x.  Launches that fuse the 2x2 max-pool keep 16 x 32
// (the pool wants a lane's two rows to be a vertical pair); they still take the strip.
#include <stdlib.h>
#include "conv.h"

#define TW 32
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

namespace {

// One 1 KiB LDS-DMA piece (see conv_x3.hip::bdma16 for the inline-asm rationale: the compiler would order every later ds_read behind
// a DMA it knows about; the kernel waits for its own pieces explicitly).
__device__ __forceinline__ void bfd_dma16(i32x4 rsrc, unsigned voff, unsigned soff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(soff), "s"(lds_dst) : "memory");
}
// wait until at most N of this wave's vector-memory operations are outstanding (they retire in issue order): everything older than the
// N youngest has landed
template <int N>
__device__ __forceinline__ void bfd_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int BN, int RPW, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void conv_bfd_kernel(const ConvArgs a) {
    constexpr int THREADS = 64 * WAVES;
    constexpr int TH = WAVES * RPW, NT = BN / 32, HW2 = TW + 2, A_PIX = (TH + 2) * HW2;
    constexpr int A_UNITS = A_PIX * 4, A_PIECES = (A_UNITS + 63) / 64, A_BYTES = A_PIECES * 1024;
    constexpr int B_UNITS = 3 * BN * 4, B_PIECES = B_UNITS / 64, B_BYTES = B_PIECES * 1024;
    constexpr int A_IT = (A_PIECES + WAVES - 1) / WAVES, B_IT = (B_PIECES + WAVES - 1) / WAVES;
    static_assert(B_UNITS % 64 == 0, "slab = whole DMA pieces");
    constexpr int NBB = 3;                                               // weight-slab ring
    extern __shared__ __attribute__((aligned(16))) char lds[];           // [B0][B1][B2][A0][A1][bias: Nout floats]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 31, hi = lane >> 5;
    const int NB = a.Nout / BN;
    const int Cin = a.C0 + a.C1, NCH = Cin >> 5, NCH0 = a.C0 >> 5;
    const int total_tiles = a.tiles_x * a.tiles_y * NB;                   // tiles_y: tile rows of the virtual strip (all images)
    const int Cs0 = a.C0;                                                 // channels per source tensor (C1 == C0 or 0)
    const int THL = a.tile_h, TWL = a.tile_w, HWL = TWL + 2;              // the launch's tile shape; halo row = TWL + 2 pixels
    const int APX = (THL + 2) * HWL, NPX = THL * TWL, VP = a.vp;          // halo pixels / tile pixels in use (<= A_PIX / TH * TW)
    const bool seam = VP % THL != 0;                                      // tiles may straddle two images (conv.h vrow_pitch)
    const bool grouped = conv_slots_grouped(THL, TWL, TH);                // slots numbered lane group by lane group (conv.h conv_tile_shape)

    const unsigned lds_base = (unsigned)__builtin_amdgcn_readfirstlane((int)(size_t)(__attribute__((address_space(3))) char*)lds);
    const unsigned ldsB_addr = lds_base, ldsA_addr = lds_base + NBB * B_BYTES;
    const unsigned long long wbase = (unsigned long long)a.wp;
    const i32x4 rsrc_w = {(int)(unsigned)wbase, (int)((unsigned)(wbase >> 32) & 0xFFFFu), (int)((size_t)3 * NCH * NB * B_BYTES), 0x00020000};
    const unsigned dma_voff = (unsigned)lane * 16u;
    constexpr unsigned OOB = 0xFFFFFFF0u;

    // ---- activation DMA: which (halo pixel, octet) lands in this lane's slot of piece wave + it*WAVES ----------------------------
    int a_hy[A_IT], a_hx[A_IT];            // halo coordinates - 1 (offsets relative to the tile origin); hy very negative: no unit
    unsigned a_oct[A_IT];
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
        const int piece = (wave + it * WAVES) % A_PIECES;                 // every wave issues A_IT pieces (a duplicate rewrites the same bytes)
        const int u = piece * 64 + lane;
        const int P = u >> 2;
        const int hr = P / HWL, hc = P - hr * HWL;
        a_hy[it] = P < APX ? hr - 1 : -(1 << 20);
        a_hx[it] = hc - 1;
        a_oct[it] = (unsigned)((u & 3) ^ ((hc >> 2) & 3)) * 16u;
    }
    unsigned a_voff[A_IT];
    int l_img = 0;
    // tile -> channel block, first strip row, first column, and the image the first row lies in (img0) with the row's offset in its pitch (vrel)
    auto decode = [&](int t, int& nb, int& img0, int& vrel, int& x0) {
        nb = t % NB;
        const int r = t / NB;
        const int ty = r / a.tiles_x, tx = r - ty * a.tiles_x;
        const int v0 = ty * THL;
        img0 = v0 / VP; vrel = v0 - img0 * VP;
        x0 = tx * TWL;
    };
    // strip row `row` relative to image img0's pitch -> row index inside the two-image window at img0 (second image: H ..), false: no such pixel row
    auto win_row = [&](int row, int& wrow) -> bool {
        const bool over = row >= VP;
        wrow = over ? row - VP + a.H : row;
        return over ? seam && row - VP < a.H : (unsigned)row < (unsigned)a.H;
    };
    auto setup_load = [&](int t) {
        int nb, vrel, x0;
        decode(t, nb, l_img, vrel, x0);
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            int wrow;
            const bool rok = win_row(vrel + a_hy[it], wrow);
            const int gx = x0 + a_hx[it];
            const bool ok = rok && gx >= 0 && gx < a.W;                  // rows of the window's second image that do not exist lie outside the descriptor
            a_voff[it] = ok ? (unsigned)(wrow * a.W + gx) * (unsigned)(Cs0 * 2) + a_oct[it] : OOB;
        }
    };
    auto dma_A = [&](int buf, int chunk) {
        if (ELD_DBG(a) & 4) return;
        const char* src = static_cast<const char*>(chunk < NCH0 ? a.in0 : a.in1);
        const int cs = chunk < NCH0 ? chunk : chunk - NCH0;
        const size_t img_bytes = (size_t)a.H * a.W * Cs0 * 2;
        const unsigned long long ab = (unsigned long long)(src + (size_t)l_img * img_bytes);
        const i32x4 rsrc_a = {(int)(unsigned)ab, (int)((unsigned)(ab >> 32) & 0xFFFFu), (int)(unsigned)(img_bytes * (size_t)(a.N - l_img < 2 ? a.N - l_img : 2)), 0x00020000};
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            const int piece = (wave + it * WAVES) % A_PIECES;            // wave-uniform
            bfd_dma16(rsrc_a, a_voff[it], (unsigned)(cs * 64), ldsA_addr + (unsigned)(buf * A_BYTES + piece * 1024));
        }
    };
    auto dma_B = [&](int buf, int nb, int chunk, int ky) {
        if (ELD_DBG(a) & 8) return;
        const unsigned soff = (unsigned)(((ky * NCH + chunk) * NB + nb) * B_BYTES);
#pragma unroll
        for (int it = 0; it < B_IT; ++it) {
            const int piece = (wave + it * WAVES) % B_PIECES;
            bfd_dma16(rsrc_w, dma_voff, soff + (unsigned)(piece * 1024), ldsB_addr + (unsigned)(buf * B_BYTES + piece * 1024));
        }
    };

    // ---- fragment addresses: per lane, per (kx, k-block); rows / taps / buffers are uniform or immediate offsets ---------------
    // slot p = (wave * RPW + r) * 32 + s(m) of the workgroup is tile pixel (p / TWL, p % TWL), s = conv_slot_of_lane; slots beyond the tile's pixel
    // count compute on pixel 0
    unsigned fx_off[RPW][3][2], fw_off[2];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        int p = (wave * RPW + r) * 32 + conv_slot_of_lane(m, grouped);
        p = p < NPX ? p : 0;
        const int tr = p / TWL, tc = p - tr * TWL;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const int hc = tc + kx;
                fx_off[r][kx][kb] = (unsigned)((tr * HWL + hc) * 64 + (((kb * 2 + hi) ^ ((hc >> 2) & 3)) * 16));
            }
    }
    // epilogue: the two pixels of each MFMA column-of-32 this lane stores in the full-line layout (slots lp and lp + 16 of column r), as
    // (tile row << 8 | tile column); 0xFFFF: the slot is beyond the tile
    unsigned ep_rc[RPW][2];
#pragma unroll
    for (int r = 0; r < RPW; ++r)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int p = (wave * RPW + r) * 32 + conv_slot_of_lane((lane & 15) + 16 * h, grouped);      // MFMA column (lane & 15) + 16 h of column-of-32 r
            const int tr = p / TWL, tc = p - tr * TWL;
            ep_rc[r][h] = p < NPX ? (unsigned)((tr << 8) | tc) : 0xFFFFu;
        }
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) fw_off[kb] = (unsigned)(m * 64 + (((kb * 2 + hi) ^ ((m >> 2) & 3)) * 16));
    const char* ldsA = lds + NBB * B_BYTES;
    const char* ldsB = lds;

    int t = xcd_block(a.xcd);
    if (t >= total_tiles) return;
    // bias -> LDS once per workgroup: a global load inside the tile loop's epilogue would be waited for with a vmcnt that also drains every
    // LDS-DMA piece in flight (the queue retires in order)
    float* lds_bias = reinterpret_cast<float*>(lds + NBB * B_BYTES + 2 * A_BYTES);
    if (a.epi == EPI_FWD) {
        for (int i = tid; i < a.Nout; i += THREADS) lds_bias[i] = a.bias[i];
    }
    __syncthreads();                             // (plain loads + ds_write: complete before the first DMA is issued)
    setup_load(t);
    int bufA = 0, bufB = 0;
    // slab of the stage `ahead` stages after (chunk, ky) of tile (nb, t_next): crosses chunk and tile boundaries; false at the end of the work
    auto dma_B_ahead = [&](int buf, int nb, int chunk, int ky, int ahead, int t_next) -> bool {
        const int q = chunk * 3 + ky + ahead;
        if (q < 3 * NCH) { dma_B(buf, nb, q / 3, q % 3); return true; }
        if (t_next < total_tiles) { const int q2 = q - 3 * NCH; dma_B(buf, t_next % NB, q2 / 3, q2 % 3); return true; }
        return false;
    };
    int young;                                   // DMA instructions this wave issued during the previous stage; < 0: wait for everything
    {
        int nb0, i0, v00, x00;
        decode(t, nb0, i0, v00, x00);
        dma_A(0, 0);
        dma_B(0, nb0, 0, 0);
        young = dma_B_ahead(1, nb0, 0, 0, 1, t + (int)gridDim.x) ? B_IT : 0;
    }
    for (;;) {
        int nb, img0, vrel, x0;
        decode(t, nb, img0, vrel, x0);
        const int t_next = t + gridDim.x;
        f32x16 acc[RPW][NT];
#pragma unroll
        for (int r = 0; r < RPW; ++r)
#pragma unroll
            for (int tt = 0; tt < NT; ++tt)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[r][tt][i] = 0.f;

        for (int chunk = 0; chunk < NCH; ++chunk) {
            const bool last_chunk = chunk + 1 >= NCH;
#pragma unroll 1
            for (int ky = 0; ky < 3; ++ky) {
                __builtin_amdgcn_s_setprio(3);
                // this wave's pieces of THIS stage's operands have landed (the younger ones, for later stages, may still fly)
                if (young == B_IT) bfd_wait_vm<B_IT>();
                else if (young == B_IT + A_IT) bfd_wait_vm<B_IT + A_IT>();
                else eld_wait_vmcnt_dyn(young);
                __syncthreads();                 // ... and everybody else's; the previous stage's fragment reads are done
                // This stage's DMA pieces (the slab two stages ahead; at ky = 0 the halo tile three stages ahead).  A piece costs its wave 100-200
                // cycles of issue time (MI355X_MICROARCH.md): waves 0-3 issue at the top of the stage, their SIMD partners 4-7 (a workgroup's
                // waves go to the SIMDs cyclically) after the first third of the MFMAs, so that one wave of every SIMD feeds the matrix pipe
                // while the other one issues: -4 % per launch against everybody issuing behind the barrier (profiles/r03_ab_notes.md).
                int issued = 0;
                auto issue_stage = [&]() {
                    int b2 = bufB + 2; if (b2 >= NBB) b2 -= NBB;
                    if (dma_B_ahead(b2, nb, chunk, ky, 2, t_next)) issued += B_IT;
                    if (ky == 0) {               // the halo tile of the next chunk / next tile has three stages to arrive
                        if (!last_chunk) { dma_A(bufA ^ 1, chunk + 1); issued += A_IT; }
                        else if (t_next < total_tiles) { setup_load(t_next); dma_A(bufA ^ 1, 0); issued += A_IT; }
                    }
                };
                const bool late = wave >= 4;
                if (!late) issue_stage();
                __builtin_amdgcn_s_setprio(0);
                const char* la = ldsA + bufA * A_BYTES + ky * (HWL * 64);
                const char* lb = ldsB + bufB * B_BYTES;
                if (!(ELD_DBG(a) & 2))
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb) {
                        uint4 fx[RPW], fw[NT];
#pragma unroll
                        for (int r = 0; r < RPW; ++r) fx[r] = *reinterpret_cast<const uint4*>(la + fx_off[r][kx][kb]);
#pragma unroll
                        for (int tt = 0; tt < NT; ++tt) fw[tt] = *reinterpret_cast<const uint4*>(lb + (kx * BN + tt * 32) * 64 + fw_off[kb]);
#pragma unroll
                        for (int r = 0; r < RPW; ++r)
#pragma unroll
                            for (int tt = 0; tt < NT; ++tt)
                                acc[r][tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fw[tt]), __builtin_bit_cast(bf16x8, fx[r]),
                                                                                     acc[r][tt], 0, 0, 0);      // D[channel][pixel]
                        if (kx == 0 && kb == 1 && late) issue_stage();
                    }
                young = issued == 0 ? -1 : issued;
                if (++bufB >= NBB) bufB = 0;
            }
            bufA ^= 1;
        }

        // ---- epilogue.  The MFMA leaves lane (m, hi) with channels 8q + 4hi .. +3 of slot m of its column-of-32 in every 32-channel block; stores
        //      (and the loads of the saved activations) use the full-line layout of conv.h bf16_line_swap: per block, instruction i of a wave covers
        //      the 64 bytes of slots 16 i .. 16 i + 15, lane l holding slot (l & 15) + 16 i, channel group bf16_line_group(l).  A slot's pixel
        //      is per-lane (ep_rc): consecutive slots are consecutive pixels of a tile row except where the row wraps.
        //      -------------------------------------------------------------------------------------------------------------------------
        if (!(ELD_DBG(a) & 1)) {
            const int lg = bf16_line_group(lane);
            // Forward: bias and max(0.2 v, v) once, in place, two values per instruction where the ISA has a packed form (the pooled copy
            // below reuses the activated values).  The whole epilogue runs with the matrix pipe idle (every wave of the workgroup reaches it
            // at the same stage), so its VALU instruction count is launch time: 2.5 instructions per value here, against 8 with a
            // software bf16 round and 14 per pooled value when the pool recomputed bias + activation.
            if (a.epi == EPI_FWD) {
                const float* lb4 = lds_bias + nb * BN + 4 * hi;
                const float sl = a.lrelu ? 0.2f : 1.0f;                  // max(1 v, v) = v: no branch inside the unrolled loops
#pragma unroll
                for (int tt = 0; tt < NT; ++tt)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 bq = *reinterpret_cast<const float4*>(lb4 + tt * 32 + 8 * q);
#pragma unroll
                        for (int r = 0; r < RPW; ++r) bias_lrelu4(acc[r][tt], 4 * q, bq, sl);
                    }
            }
            // (Pairing adjacent blocks into whole 128-byte lines per store instruction -- 8 pixels x 128 B instead of 16 pixels x 64 B -- was
            // measured and dropped: 2 % slower here, neutral on conv_bfw; profiles/r03_ab_notes.md.)
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                // this lane's two pixels of column r: NHW pixel index (0 where there is none: loads re-read a valid pixel, stores are predicated)
                size_t pix[2];
                bool pok[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int row = vrel + (int)(ep_rc[r][h] >> 8), x = x0 + (int)(ep_rc[r][h] & 255u);
                    const bool over = row >= VP;
                    const int y = over ? row - VP : row, img = img0 + (over ? 1 : 0);
                    pok[h] = ep_rc[r][h] != 0xFFFFu && y < a.H && img < a.N && x < a.W;
                    pix[h] = pok[h] ? (size_t)(img * a.H + y) * a.W + x : 0;
                }
#pragma unroll
                for (int tt = 0; tt < NT; ++tt) {
                    const int nb32 = nb * BN + tt * 32;
                    float4 v[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = make_float4(acc[r][tt][4 * q], acc[r][tt][4 * q + 1], acc[r][tt][4 * q + 2], acc[r][tt][4 * q + 3]);
                    bf16_t* dst;                                         // channel 0 of this lane's group in pixel 0 of the destination tensor
                    int C;
                    if (a.epi == EPI_FWD) {
                        C = a.Nout;
                        dst = static_cast<bf16_t*>(a.out0) + nb32 + 8 * lg;
                    } else {
                        const bool lo = nb32 < a.split;
                        C = lo ? a.split : a.Nout - a.split;
                        const int cb = lo ? nb32 : nb32 - a.split;
                        dst = static_cast<bf16_t*>(lo ? a.out0 : a.out1) + cb + 8 * lg;
                        const bf16_t* act = static_cast<const bf16_t*>(lo ? a.act0 : a.act1);
                        if (act != nullptr) {
                            const bf16_t* arow = act + cb + 8 * lg;
                            const uint4 a0 = *reinterpret_cast<const uint4*>(arow + pix[0] * C);
                            const uint4 a1 = *reinterpret_cast<const uint4*>(arow + pix[1] * C);
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
                    bf16_line_swap(pk, s0, s1);                          // every lane takes part; only the stores are predicated
                    if (pok[0]) *reinterpret_cast<uint4*>(dst + pix[0] * C) = s0;
                    if (pok[1]) *reinterpret_cast<uint4*>(dst + pix[1] * C) = s1;
                }
            }
            // fused nn.MaxPool2d(2) (Unet.py:51-63): vertical pair in the lane's own rows, horizontal pair in lane ^ 1, on the activated
            // fp32 values (max commutes with the monotone bf16 rounding, so this equals pooling the stored tensor).  Pooled launches run
            // 16 x 32 tiles (launcher): slot m of column r is pixel (wave * RPW + r, m) of the tile, and strip rows keep their parity.
            if (a.epi == EPI_FWD && a.pool_out != nullptr) {
                const int Hp = a.H >> 1, Wp = a.W >> 1;
                const int x = x0 + m;
                const bool xok = x < a.W;
#pragma unroll
                for (int tt = 0; tt < NT; ++tt) {
#pragma unroll
                    for (int rp = 0; rp < RPW / 2; ++rp) {
                        const int row = vrel + wave * RPW + 2 * rp;     // wave-uniform
                        const bool over = row >= VP;
                        const int y = over ? row - VP : row, img = img0 + (over ? 1 : 0);
                        if (y >= a.H || img >= a.N) continue;
                        uint2 pk[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            float u[4];
#pragma unroll
                            for (int j = 0; j < 4; ++j) u[j] = fmax_lane_xor1(fmax_raw(acc[2 * rp][tt][4 * q + j], acc[2 * rp + 1][tt][4 * q + j]));
                            pk[q] = pack_bf4(make_float4(u[0], u[1], u[2], u[3]));
                        }
                        // (every lane takes part in the exchanges; even pixels inside the image store)
                        const uint4 w0 = bf16_pair_swap(pk[0], pk[1]), w1 = bf16_pair_swap(pk[2], pk[3]);
                        if (xok && !(x & 1)) {
                            bf16_t* dp = static_cast<bf16_t*>(a.pool_out) + ((size_t)(img * Hp + (y >> 1)) * Wp + (x >> 1)) * a.Nout + nb * BN + tt * 32 + 8 * hi;
                            *reinterpret_cast<uint4*>(dp) = w0;
                            *reinterpret_cast<uint4*>(dp + 16) = w1;
                        }
                    }
                }
            }
        }
        // The next stage waits with the same count as if there had been no epilogue ("at most `young` operations outstanding").  The epilogue's
        // stores are the youngest entries of the queue, so this also waits for all but the last few of them -- measured (same-box A/B,
        // profiles/r03_ab_notes.md) that is no slower than counting the stores exactly and letting them fly, it needs no assumption about loads
        // and stores retiring in one common order, and unlike the old full drain (vmcnt(0)) it never waits for the youngest slab pieces.
        young = young > 0 ? young : 0;
        if (t_next >= total_tiles) break;
        t = t_next;
    }
}

template <int BN, int RPW, int WAVES>
int launch_bfd(ConvArgs a, hipStream_t st) {
    a.xcd = eld_xcd_mask() & XCD_BF16;
    constexpr int TH = WAVES * RPW;
    conv_tile_shape(a.N, a.H, a.W, TH, a.pool_out != nullptr, a.tile_h, a.tile_w);
    a.vp = vrow_pitch(a.N, a.H, a.tile_h);
    a.tiles_x = (a.W + a.tile_w - 1) / a.tile_w;
    a.tiles_y = (vrow_extent(a.N, a.H, a.vp) + a.tile_h - 1) / a.tile_h;
    constexpr size_t A_BYTES = (size_t)(((TH + 2) * (TW + 2) * 4 + 63) / 64) * 1024, B_BYTES = (size_t)3 * BN * 64;
    if (a.Nout > 1024) return ELD_ENOTSUP;
    const size_t lds_bytes = 2 * A_BYTES + 3 * B_BYTES + 4096;      // + bias (up to 1024 floats)
    const long long tiles = (long long)a.tiles_x * a.tiles_y * (a.Nout / BN);
    if (tiles <= 0) return 0;
    if (tiles > 0x7fffffffLL) return ELD_ENOTSUP;
    auto kern = conv_bfd_kernel<BN, RPW, WAVES>;
    static EldAttrOnce once;
    { const int rc = once.ensure(kern, lds_bytes); if (rc) return rc; }
    long long grid = (long long)eld_num_cus();
    if (grid > tiles) grid = tiles;
    ELD_LAUNCH(kern, dim3((unsigned)grid), dim3(64 * WAVES), lds_bytes, st, a);
    ELD_LAUNCH_CHECK();
    return 0;
}

}  // namespace

// Layers the DMA kernel takes: bf16 3x3 with GEMM N (Nout) a multiple of 64 and K a multiple of 32, on a tile domain (N images of H x W) that
// gives every CU one of its 16-row tiles; returns the slab's channel-block width BN (what the pack kernel must lay the weights out for), or 0 for
// launches that stay on conv_igemm_kernel<bf16_t> (32-channel layers; small problems, where its 8-row tiles with two 4-wave workgroups per CU
// spread the work over more CUs).
int bfd_slab_bn(int Nout, int K, int N, int H, int W) {
    if (bfs_takes(Nout, K, N, H, W)) return 32;      // conv_bfs.hip: the same slab layout at BN = 32
    if (debug_kernel_mask(-1) & 4) return 0;         // test hook: everything else back on conv_igemm_kernel<bf16>
    if (K % 32 || Nout % 64) return 0;
    const long long px_tiles = conv_tile_count(N, H, W, 16, false);
    const int cus = eld_num_cus();
    if (Nout % 128 == 0 && px_tiles * (Nout / 128) >= cus) return 128;
    return px_tiles * (Nout / 64) >= cus ? 64 : 0;
}

// a: bf16 CONV_3X3 arguments already validated by launch_conv; weights in slab layout
int launch_conv_bfd(const ConvArgs& a, hipStream_t st) {
    if ((size_t)a.H * a.W * a.C0 * 2 * (a.N > 1 ? 2 : 1) >= 0xFFFFFFF0ull) return ELD_ENOTSUP;      // conv_bfd_kernel addresses a two-image window (virtual rows)
    if (a.pool_out && (a.epi != EPI_FWD || (a.H & 1) || (a.W & 1))) return ELD_EINVAL;
    const int bn = bfd_slab_bn(a.Nout, a.C0 + a.C1, a.N, a.H, a.W);
    if (bn == 32) return launch_conv_bfs(a, st);
    if (bn == 128) return launch_bfd<128, 2, 8>(a, st);
    if (bn == 64) return bfw_takes(a) ? launch_conv_bfw(a, st) : launch_bfd<64, 2, 8>(a, st);
    return ELD_ENOTSUP;
}
