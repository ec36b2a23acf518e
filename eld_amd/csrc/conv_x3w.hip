// conv_x3w.hip -- fp32 3x3 convolution for the 32-OUTPUT-CHANNEL full-resolution layers (conv1_2, conv9_1, conv9_2 forward; the backward-data of conv9_2,
// conv1_2 and conv2_1), three-piece scheme of conv_x3.hip, with the waves of a workgroup SPECIALISED (round 6).  Same contract and the same bits as
// conv_x3_kernel<32, 4> (models/arch/Unet.py:49-51,83-88 and their autograd backward-data).
//
// Why (profiles/r06_ab_notes.md): in conv_x3_kernel<32, 4> every wave does everything in turn -- wait for its halo loads, cut them into bf16 pieces, store
// them to LDS, barrier, 72 MFMAs, ... epilogue -- and a tile's memory side (78 KB in, 64-80 KB out) takes about as long as its matrix side (13.8 K pipe
// cycles per wave); with two such workgroups per CU the two sides ran almost back to back (matrix pipe 50 % busy), because a workgroup can hold only ONE
// 16-channel chunk of loads in flight (40 staging VGPRs at a full register file).  Here ONE 16-wave workgroup owns the CU (the shipped configuration; the template
// also builds 4 + 4 and 8 + 4 waves, profiles/r06_ab_notes.md):
//   * waves 0-7 (two per SIMD, two pixel rows each) are CONSUMERS: fragment reads + MFMAs (x3_stage_blocks) and the tile's epilogue, nothing else;
//   * waves 8-15 (two more per SIMD; 122 VGPRs per wave) are PRODUCERS: they run the chunk sequence of the workgroup's tiles AHEAD of the consumers -- halo loads of chunk g+3
//     into one of two register sets (two chunks in flight per CU instead of one: the producers have the registers the consumers' accumulators do not
//     leave the others) and the exact three-piece cut of chunk g+1 into the OTHER half of a double-buffered LDS tile while the consumers multiply chunk g.
//     Seven of them stage the halo; the eighth moves the next stage's pre-split weight slab (conv_x3d_kernel's slab layout, pack kernel) by LDS-DMA and has
//     nothing else in flight (vmcnt retires in order: whoever waits for slab pieces also waits for every older load or store of its own).
// VALU / LDS-write / VMEM work of a SIMD's producer wave and the MFMA stream of its consumer wave are different pipes: they overlap by construction instead of
// by the luck of two workgroups' phases.  One workgroup barrier per stage (kernel row) hands the finished halves over.
//
// LDS: 2 x 68,544 B (halo tile of one 16-channel chunk: 18 x 34 pixels x [3 pieces][16 bf16] + pad, conv_x3_kernel's image) + 2 x 11,264 B (slab ring)
//      = 159,616 B of the CU's 163,840.
#include <stdlib.h>
#include <type_traits>
#include "conv_x3_dev.h"

namespace {

template <int N>
__device__ __forceinline__ void w_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// dev tool (ELD_DEV_TOOLS builds, eld_debug_conv_prof): (tag, s_memtime) pairs of the first consumer / halo / slab wave of the first 8 workgroups, 1024 per wave
struct WProf {
    unsigned long long* p; int n;
    __device__ __forceinline__ void operator()(unsigned tag) {
        if (p != nullptr && n < 1024) { p[n] = ((unsigned long long)tag << 56) | (__builtin_readcyclecounter() & 0x00FFFFFFFFFFFFFFull); ++n; }
    }
};

constexpr int W_BN = 32, W_TH = 16, W_CK = 16;          // (consumer waves x rows per consumer wave = W_TH: 4 x 4 or 8 x 2, template parameters)
constexpr int W_APIX = (W_TH + 2) * (TW + 2);                       // 612 halo pixels
constexpr int W_AWORDS = W_APIX * PX;                               // 17,136 words = 68,544 B
constexpr int W_BROWS = 3 * W_BN;
constexpr int W_BWORDS = (W_BROWS * PX * 4 + 1023) / 1024 * 256;    // 2,816 words = 11,264 B (x3_slab_stride(32))
constexpr int W_BPIECES = W_BWORDS * 4 / 1024;                      // 11
constexpr int W_AUNITS = W_APIX * 4;

// SLABW: the last producer wave moves the weight slabs and the other three stage the halo (13 units per thread and chunk: 216 VGPRs, so only with two waves per
// SIMD = 4 consumer waves); else all four producer waves stage the halo (10 units) AND move the slabs, with a counted wait for their pieces.
template <int W_CW, int W_RPW, bool SLABW, bool PRIO, int W_PW = 4, bool PAIR = false>
__global__ __launch_bounds__(64 * (W_CW + W_PW), (W_CW + W_PW) / 4) void conv_x3w_kernel(const ConvArgs a) {
    static_assert(W_CW * W_RPW == W_TH, "tile rows");
    static_assert(!PAIR || SLABW, "paired halo loads: the halo waves wait for nothing but their own loads");
    constexpr int W_HW = SLABW ? W_PW - 1 : W_PW;                     // producer waves that stage the halo
    constexpr int W_PTHREADS = 64 * W_HW;
    constexpr int W_AIT = (W_AUNITS + W_PTHREADS - 1) / W_PTHREADS;   // sixteen-byte units per halo thread and chunk: 13 / 10
    constexpr int W_T0 = (W_AIT + 2) / 3, W_T1 = W_T0 + (W_AIT - W_T0 + 1) / 2;      // a chunk's units are cut in three parts, one per stage: 13 -> 5 4 4, 10 -> 4 3 3, 5 -> 2 2 1
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* ldsB = lds;                                                // [2][W_BWORDS]: the DMA destinations stay at low LDS addresses
    float* ldsA = lds + 2 * W_BWORDS;                                 // [2][W_AWORDS]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool consumer = wave < W_CW;
    const int m = lane & 31, hi = lane >> 5;
    const int Cin = a.C0 + a.C1, NCH = Cin / W_CK;                    // 2 or 4 chunks per tile (launcher: NCH even, C0 % 32 == 0)
    const int tiles_img = a.tiles_x * a.tiles_y;
    const int total_tiles = tiles_img * a.N;
    const int first = xcd_block(a.xcd), stride = gridDim.x;
    if (first >= total_tiles) return;
    const int my_tiles = (total_tiles - first + stride - 1) / stride;
    const int n_chunks = my_tiles * NCH;                              // the workgroup's chunk sequence g = 0 .. n_chunks - 1: tile first + (g / NCH) * stride, chunk g % NCH

    auto decode = [&](int t, int& img, int& y0, int& x0) {
        img = t / tiles_img;
        const int r = t - img * tiles_img;
        int ty, tx;
        band_tile(r, a.tiles_x, a.tiles_y, a.band, ty, tx);
        y0 = ty * W_TH; x0 = tx * TW;
    };

    unsigned long long* prof_base = ELD_PROF(a);
    const int prole = wave == 0 ? 0 : (wave == W_CW ? 1 : (wave == W_CW + W_PW - 1 ? 2 : -1));
    WProf prof = {(prof_base != nullptr && blockIdx.x < 8 && prole >= 0 && lane == 0) ? prof_base + ((size_t)blockIdx.x * 3 + prole) * 1024 : nullptr, 0};
    const unsigned ldsB_addr = (unsigned)__builtin_amdgcn_readfirstlane((int)(size_t)(__attribute__((address_space(3))) float*)ldsB);
    const unsigned long long wbase = (unsigned long long)a.wp;
    const i32x4 rsrc_w = {(int)(unsigned)wbase, (int)((unsigned)(wbase >> 32) & 0xFFFFu), (int)((size_t)3 * NCH * W_BWORDS * 4), 0x00020000};
    using I0_ = std::integral_constant<int, 0>;
    using I1_ = std::integral_constant<int, 1>;
    using IA_ = std::integral_constant<int, W_T0>;
    using IB_ = std::integral_constant<int, W_T1>;
    using IE_ = std::integral_constant<int, W_AIT>;

    // slab (chunk c, kernel row ky) -> ring buffer `buf`, all eleven 1 KiB pieces by the ONE slab wave.  vmcnt retires in order, so whoever waits for slab pieces
    // also waits for every older vector-memory operation of its own: a consumer would retire its epilogue's stores (6.5 K cycles per tile, measured as -2.4 % when
    // the pieces moved to the producers), a halo wave its loads of the previous stage (the stage time then follows the memory latency).  The slab wave has nothing
    // else in flight.
    auto dma_slab = [&](int buf, int c, int ky) {
        const unsigned soff = (unsigned)((ky * NCH + c) * (W_BWORDS * 4));
        constexpr int MOVERS = SLABW ? 1 : W_PW;
        const int first_piece = SLABW ? 0 : wave - W_CW;
#pragma unroll
        for (int it = 0; it < (W_BPIECES + MOVERS - 1) / MOVERS; ++it) {
            const int piece = first_piece + it * MOVERS;              // wave-uniform
            if (piece < W_BPIECES) bdma16(rsrc_w, (unsigned)lane * 16u, soff + (unsigned)(piece * 1024), ldsB_addr + (unsigned)(buf * W_BWORDS * 4 + piece * 1024));
        }
    };

    // The two roles are two separate loops over the same stage sequence (prologue barrier, then one barrier per (tile, chunk, kernel row)): their registers
    // -- the producers' two staging sets, the consumers' accumulators and fragments -- never compete for one allocation.
    if (SLABW && wave == W_CW + W_HW) {
        // ======================================================================================================================================
        // SLAB wave: the next stage's weight slab, one stage ahead
        // ======================================================================================================================================
        if constexpr (PRIO) __builtin_amdgcn_s_setprio(3);
        dma_slab(0, 0, 0);
        dma_wait();
        __syncthreads();
        int g = 0, sb = 0;
        for (int k = 0; k < my_tiles; ++k)
            for (int c = 0; c < NCH; ++c) {
                for (int ky = 0; ky < 3; ++ky) {
                    int c1 = c, ky1 = ky + 1;
                    if (ky1 == 3) { ky1 = 0; c1 = c + 1 < NCH ? c + 1 : 0; }
                    prof(20);
                    if (g + 1 < n_chunks || ky < 2) dma_slab(sb ^ 1, c1, ky1);      // (the other ring buffer: everybody finished reading it at the last barrier)
                    prof(21);
                    dma_wait();
                    prof(22);
                    __syncthreads();
                    sb ^= 1;
                }
                ++g;
            }
        return;
    }
    if (!consumer) {
        // ======================================================================================================================================
        // HALO waves: halo loads two chunks ahead, the three-piece cut one chunk ahead
        // ======================================================================================================================================
        // The producers share their SIMDs with the consumers' MFMA / ds_read streams and are the YOUNGER waves: VALU issue goes to the older wave first
        // (MI355X_MICROARCH.md, two waves per SIMD), so without a priority their short cut / store bursts queue behind it and every stage ends late.
        if constexpr (PRIO) __builtin_amdgcn_s_setprio(3);
        const int ptid = tid - 64 * W_CW;                             // 0 .. 255
        constexpr unsigned OOB = 0xFFFFFFF0u;
        float4 ra[2][W_AIT];                                          // register set s holds the halo of the chunks g with (g & 1) == s (NCH is even: s = chunk & 1)
        unsigned a_voff[2][W_AIT];
        int s_img[2] = {0, 0}, s_tile[2] = {-1, -1};
        // addresses of set s for tile t (a chunk's units: unit u = ptid + it * 256 -> halo pixel stage_row(u >> 2), 16-byte part u & 3)
        auto setup_set = [&](auto S, int t) {
            constexpr int s = decltype(S)::value;
            int y0, x0;
            decode(t, s_img[s], y0, x0);
            s_tile[s] = t;
#pragma unroll
            for (int it = 0; it < W_AIT; ++it) {
                const int u = ptid + it * W_PTHREADS;
                const int hp = stage_row(u >> 2, W_APIX), part = u & 3;
                const int hy = hp / (TW + 2), hx = hp - hy * (TW + 2);
                const int gy = y0 + hy - 1, gx = x0 + hx - 1;
                const bool ok = u < W_AUNITS && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
                a_voff[s][it] = ok ? (unsigned)(gy * a.W + gx) * (unsigned)(a.C0 * 4) + (unsigned)part * 16u : OOB;
            }
        };
        // loads of units [I0, I1) of chunk g into its set (no-op past the end of the sequence)
        auto load_part = [&](auto S, auto I0, auto I1, int g) {
            constexpr int s = decltype(S)::value;
            if (g >= n_chunks) return;
            const int t = first + (g / NCH) * stride, c = g % NCH;
            if (t != s_tile[s]) setup_set(S, t);
            const int c0 = c * W_CK;
            const char* src = static_cast<const char*>(c0 < a.C0 ? a.in0 : a.in1);
            const int cs = c0 < a.C0 ? c0 : c0 - a.C0;
            const size_t img_bytes = (size_t)a.H * a.W * a.C0 * 4;
            const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc((void*)(src + (size_t)s_img[s] * img_bytes), 0, (int)img_bytes, 0x00020000);
#pragma unroll
            for (int it = decltype(I0)::value; it < decltype(I1)::value; ++it)
                ra[s][it] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, (int)a_voff[s][it], cs * 4, 0));
        };
        // cut units [I0, I1) of the chunk held by set s into half `s` of the LDS tile
        auto cut_part = [&](auto S, auto I0, auto I1) {
            constexpr int s = decltype(S)::value;
#pragma unroll
            for (int it = decltype(I0)::value; it < decltype(I1)::value; ++it) {
                const int u = ptid + it * W_PTHREADS;
                if (u < W_AUNITS) split_store(ldsA + s * W_AWORDS + stage_row(u >> 2, W_APIX) * PX + (u & 3) * 2, ra[s][it]);
            }
        };
        // prologue: (the first slab,) chunk 0 cut into half 0; chunks 1 and 2 in flight
        if constexpr (!SLABW) dma_slab(0, 0, 0);                      // (older than every load below: landed when chunk 0's registers are)
        load_part(I0_{}, I0_{}, IE_{}, 0);
        load_part(I1_{}, I0_{}, IE_{}, 1);
        cut_part(I0_{}, I0_{}, IE_{});
        if constexpr (!PAIR) load_part(I0_{}, I0_{}, IE_{}, 2);
        __syncthreads();
        int g = 0, sb = 0;
        for (int k = 0; k < my_tiles; ++k) {
            for (int cp = 0; cp < NCH; cp += 2) {
                // two chunks per trip so that register-set / LDS-half indices are compile-time constants: chunk g (even) is consumed from half 0, g + 1 from half 1
                auto chunk = [&](auto SN, int c) {                     // SN: the set / half PRODUCED while the other half is consumed; c: the chunk being consumed
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky) {
                        if constexpr (!SLABW) {
                            // the next stage's slab first: its pieces are then OLDER than this stage's halo loads, which the counted wait below leaves in flight
                            int c1 = c, ky1 = ky + 1;
                            if (ky1 == 3) { ky1 = 0; c1 = c + 1 < NCH ? c + 1 : 0; }
                            if (g + 1 < n_chunks || ky < 2) dma_slab(sb ^ 1, c1, ky1);      // (the other ring buffer: everybody finished reading it at the last barrier)
                        }
                        // chunk g + 1: this stage's third of its units into LDS, then the same third of chunk g + 3 into the freed registers
                        prof(10);
                        if constexpr (PAIR) {
                            // Paired loads: chunks 2j and 2j + 1 of a tile are the two 64-byte halves of the same 128-byte pixel records (32 channels x 4 B), and
                            // the L2 fetches whole lines.  Issued a chunk period apart (the schedule below), the second half finds its line evicted again
                            // (FETCH_SIZE: 5.72 GB read per launch against 4.07 GB of halo tiles).  Here BOTH halves of a third's pixels are
                            // requested back to back, in the stages of the EVEN chunk: the odd chunk g + 3 into the registers the cut of chunk g + 1 has just
                            // freed, the even chunk g + 2 into the other set (free since chunk g was cut); the odd chunk's stages only cut.  An even chunk's
                            // units are cut three stages after their loads were issued, an odd chunk's six.  Measured (profiles/r06_ab_notes.md): 4.09 GB per launch,
                            // but the launch takes 2.7 % longer (all of a pair's loads in three stages instead of six) and the step is unchanged: opt-in (ELD_X3W=7).
                            constexpr bool even = decltype(SN)::value == 1;          // (the set being cut holds an odd chunk <=> the chunk being consumed is even)
                            using SO = std::integral_constant<int, 1 - decltype(SN)::value>;
                            if (g + 1 < n_chunks) {
                                if (ky == 0) cut_part(SN, I0_{}, IA_{}); else if (ky == 1) cut_part(SN, IA_{}, IB_{}); else cut_part(SN, IB_{}, IE_{});
                            }
                            prof(11);
                            if constexpr (even) {
                                if (ky == 0) { load_part(SO{}, I0_{}, IA_{}, g + 2); load_part(SN, I0_{}, IA_{}, g + 3); }
                                else if (ky == 1) { load_part(SO{}, IA_{}, IB_{}, g + 2); load_part(SN, IA_{}, IB_{}, g + 3); }
                                else { load_part(SO{}, IB_{}, IE_{}, g + 2); load_part(SN, IB_{}, IE_{}, g + 3); }
                            }
                        } else
                        if (g + 1 < n_chunks) {
                            if (ky == 0) { cut_part(SN, I0_{}, IA_{}); prof(11); load_part(SN, I0_{}, IA_{}, g + 3); }
                            else if (ky == 1) { cut_part(SN, IA_{}, IB_{}); prof(11); load_part(SN, IA_{}, IB_{}, g + 3); }
                            else { cut_part(SN, IB_{}, IE_{}); prof(11); load_part(SN, IB_{}, IE_{}, g + 3); }
                        }
                        prof(12);
                        if constexpr (!SLABW) {
                            if (g + 3 < n_chunks) {                    // this wave's slab pieces have landed (and, vmcnt being in order, every older load); the loads just issued may fly
                                if (ky == 0) w_wait_vm<W_T0>(); else if (ky == 1) w_wait_vm<W_T1 - W_T0>(); else w_wait_vm<W_AIT - W_T1>();
                            } else dma_wait();
                        }
                        prof(13);
                        __syncthreads();                              // this third of half SN (/ the next slab) is in LDS (all three thirds by the chunk's last stage)
                        sb ^= 1;
                    }
                    ++g;
                };
                chunk(I1_{}, cp);
                chunk(I0_{}, cp + 1);
            }
        }
        return;
    }

    // ==========================================================================================================================================
    // CONSUMER waves: fragment reads, MFMAs, the epilogue -- no vector-memory operation inside a tile
    // ==========================================================================================================================================
    __syncthreads();
    f32x16 acc[W_RPW][1];
    int sb = 0;                                                       // slab ring position of the current stage
    int g = 0;
    for (int k = 0; k < my_tiles; ++k) {
        const int t = first + k * stride;
#pragma unroll
        for (int r = 0; r < W_RPW; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[r][0][i] = 0.f;
        for (int c = 0; c < NCH; ++c) {
            const float* la = ldsA + (c & 1) * W_AWORDS;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const float* lb = ldsB + sb * W_BWORDS;
                prof(1);
                x3_stage_blocks<W_RPW, 1>(acc,
                    [&](int kx, int r, uint4 (&X)[3]) {
                        const float* p = la + ((wave * W_RPW + r + ky) * (TW + 2) + m + kx) * PX + hi * 4;
#pragma unroll
                        for (int pc = 0; pc < 3; ++pc) X[pc] = *reinterpret_cast<const uint4*>(p + pc * 8);
                    },
                    [&](int kx, int tt, uint4 (&Wt)[3]) {
                        const float* p = lb + (kx * W_BN + m) * PX + hi * 4;
#pragma unroll
                        for (int pc = 0; pc < 3; ++pc) Wt[pc] = *reinterpret_cast<const uint4*>(p + pc * 8);
                    });
                prof(2);
                __syncthreads();                                      // the producers' third of the other halo half and the next slab are in LDS
                sb ^= 1;
            }
            ++g;
        }

        prof(3);
        // ---- epilogue (consumer waves = the four waves of conv_x3_kernel<32, 4>: lane (m, hi) of wave w owns pixel x0 + m of rows y0 + 4 w .. + 3 and channels
        //      8q + 4hi .. + 3): identical to conv_x3_kernel's ----------------------------------------------------------------------------------------------
        int img, y0, x0;
        decode(t, img, y0, x0);
        constexpr int RPW = W_RPW, NT = 1, BN = W_BN;
        const int nb = 0;
        {
            const int x = x0 + m;
            const bool xok = x < a.W;
            if (a.epi == EPI_FWD) {
                const float sl = a.lrelu ? 0.2f : 1.0f;                  // max(1 v, v) = v
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 bq = *reinterpret_cast<const float4*>(a.bias + 4 * hi + 8 * q);
#pragma unroll
                    for (int r = 0; r < RPW; ++r) bias_lrelu4(acc[r][0], 4 * q, bq, sl);
                }
                if (a.codes_out != nullptr && xok) {                     // slope codes of the finished values, for the backward-data epilogue that will want them
#pragma unroll
                    for (int r = 0; r < RPW; ++r) {
                        const int y = y0 + wave * RPW + r;
                        if (y >= a.H) continue;
                        const size_t pix = (size_t)(img * a.H + y) * a.W + x;
                        a.codes_out[pix * 2 + hi] = slope_codes16(acc[r][0]);
                    }
                }
            }
            unsigned cw[RPW];                                          // EPI_GRAD with slope codes: the whole tile's words, one load each, issued together
            if (a.epi != EPI_FWD && a.codes0 != nullptr) {
#pragma unroll
                for (int r = 0; r < RPW; ++r) {
                    const int y = y0 + wave * RPW + r;
                    const size_t pix = (size_t)(img * a.H + (y < a.H ? y : 0)) * a.W + (xok ? x : 0);
                    cw[r] = a.codes0[pix * 2 + hi];
                }
            }
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                const int y = y0 + wave * RPW + r;                      // wave-uniform
                const bool yok = y < a.H;
                const size_t rowpix = (size_t)(img * a.H + (yok ? y : 0)) * a.W;
                const size_t pix = rowpix + (xok ? x : 0);
                float4 v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = make_float4(acc[r][0][4 * q], acc[r][0][4 * q + 1], acc[r][0][4 * q + 2], acc[r][0][4 * q + 3]);
                float* blk = static_cast<float*>(a.out0) + (rowpix + x0) * BN;
                if (a.epi != EPI_FWD) {
                    const float* act = static_cast<const float*>(a.act0);
                    if (a.codes0 != nullptr) {
                        const unsigned w = cw[r];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            v[q].x *= slope_of_code(w, 4 * q); v[q].y *= slope_of_code(w, 4 * q + 1);
                            v[q].z *= slope_of_code(w, 4 * q + 2); v[q].w *= slope_of_code(w, 4 * q + 3);
                        }
                    } else if (act != nullptr) {
                        float4 s[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) s[q] = *reinterpret_cast<const float4*>(act + pix * BN + 4 * hi + 8 * q);
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            v[q].x *= lrelu_slope(s[q].x); v[q].y *= lrelu_slope(s[q].y);
                            v[q].z *= lrelu_slope(s[q].z); v[q].w *= lrelu_slope(s[q].w);
                        }
                    }
                }
                f32_line_store(v, blk, (size_t)BN, lane, yok, a.W - x0);
            }
        }
        if (a.epi == EPI_FWD && a.pool_out != nullptr) {
            int img_p[RPW / 2], y_p[RPW / 2];
#pragma unroll
            for (int rp = 0; rp < RPW / 2; ++rp) { img_p[rp] = img; y_p[rp] = y0 + wave * RPW + 2 * rp; }
            pool_epilogue<RPW, NT, BN>(a, acc, img_p, nb, y_p, x0 + m, hi);
        }
        prof(4);
    }
}

}  // namespace

// Layers this kernel takes: fp32 three-piece 3x3 with exactly 32 output channels (GEMM N) and 32 or 64 input channels (one tensor, or the virtual concat of two
// 32-channel tensors), a single 32-channel output tensor, on a tile domain that gives every CU a tile; weights in conv_x3d_kernel's slab layout at BN = 32.
// ELD_X3W=0 keeps these layers on conv_x3_kernel<32, 4>.
// ELD_X3W: 0 = off (conv_x3_kernel<32, 4>), 1 = 4 consumer waves x 4 rows + 3 halo waves + 1 slab wave, 2 = 8 consumer waves x 2 rows + 4 producer waves (two MFMA
// waves per SIMD cover each other's fragment-read latency), 3 = as 2 with the producers at s_setprio 3, 4 = 8 consumer + 8 producer waves (four waves per SIMD), 5 = as 4 with one of the producer waves moving the slabs and nothing else (the default), 7 = as 5 with the halo loads of a chunk pair issued together (opt-in: -28 % of the launch's HBM reads, +2.7 % of its time)
static int x3w_mode() {
    static const int on = [] { const char* e = getenv("ELD_X3W"); return e ? atoi(e) : 5; }();
    return on;
}
bool x3w_enabled() { return x3w_mode() != 0; }
bool x3w_takes(const ConvArgs& a) {
    if (!x3w_enabled() || a.Nout != 32 || a.dtype != DT_F32) return false;
    const int Cin = a.C0 + a.C1;
    if ((Cin != 32 && Cin != 64) || a.C0 % 32) return false;
    if (a.epi == EPI_GRAD && (a.split != 32 || a.out1 != nullptr || a.act1 != nullptr || a.codes1 != nullptr)) return false;
    if (a.epi != EPI_FWD && a.epi != EPI_GRAD) return false;
    const long long tiles = (long long)((a.W + TW - 1) / TW) * ((a.H + W_TH - 1) / W_TH) * a.N;
    return tiles >= eld_num_cus();
}

static unsigned long long* g_x3w_prof = nullptr;
void conv_x3_set_prof(unsigned long long* buf) { g_x3w_prof = buf; }      // dev tool (eld_debug_conv_prof): see WProf

int launch_conv_x3w(const ConvArgs& a_in, hipStream_t st) {
    ConvArgs a = a_in;
    a.xcd = eld_xcd_mask() & XCD_X3W;
    a.band = eld_tile_band();
    a.prof = ELD_DEV_TOOLS ? g_x3w_prof : nullptr;
    a.tiles_x = (a.W + TW - 1) / TW;
    a.tiles_y = (a.H + W_TH - 1) / W_TH;
    const long long tiles = (long long)a.tiles_x * a.tiles_y * a.N;
    if (tiles <= 0) return 0;
    if (tiles > 0x3fffffffLL) return ELD_ENOTSUP;
    constexpr size_t lds_bytes = (size_t)(2 * W_BWORDS + 2 * W_AWORDS) * sizeof(float);
    long long grid = (long long)eld_num_cus();
    if (grid > tiles) grid = tiles;
    const int mode = x3w_mode();
    if (mode == 7) {
        static EldAttrOnce once;
        { const int rc = once.ensure(conv_x3w_kernel<8, 2, true, false, 8, true>, lds_bytes); if (rc) return rc; }
        ELD_LAUNCH((conv_x3w_kernel<8, 2, true, false, 8, true>), dim3((unsigned)grid), dim3(1024), lds_bytes, st, a);
    } else if (mode == 5) {
        static EldAttrOnce once;
        { const int rc = once.ensure(conv_x3w_kernel<8, 2, true, false, 8>, lds_bytes); if (rc) return rc; }
        ELD_LAUNCH((conv_x3w_kernel<8, 2, true, false, 8>), dim3((unsigned)grid), dim3(1024), lds_bytes, st, a);
    } else if (mode == 4) {
        static EldAttrOnce once;
        { const int rc = once.ensure(conv_x3w_kernel<8, 2, false, false, 8>, lds_bytes); if (rc) return rc; }
        ELD_LAUNCH((conv_x3w_kernel<8, 2, false, false, 8>), dim3((unsigned)grid), dim3(1024), lds_bytes, st, a);
    } else if (mode == 1) {
        static EldAttrOnce once;
        { const int rc = once.ensure(conv_x3w_kernel<4, 4, true, false>, lds_bytes); if (rc) return rc; }
        ELD_LAUNCH((conv_x3w_kernel<4, 4, true, false>), dim3((unsigned)grid), dim3(512), lds_bytes, st, a);
    } else if (mode == 3) {
        static EldAttrOnce once;
        { const int rc = once.ensure(conv_x3w_kernel<8, 2, false, true>, lds_bytes); if (rc) return rc; }
        ELD_LAUNCH((conv_x3w_kernel<8, 2, false, true>), dim3((unsigned)grid), dim3(768), lds_bytes, st, a);
    } else {
        static EldAttrOnce once;
        { const int rc = once.ensure(conv_x3w_kernel<8, 2, false, false>, lds_bytes); if (rc) return rc; }
        ELD_LAUNCH((conv_x3w_kernel<8, 2, false, false>), dim3((unsigned)grid), dim3(768), lds_bytes, st, a);
    }
    ELD_LAUNCH_CHECK();
    return 0;
}
