// conv_bfs.hip -- bf16 3x3 convolution for the 32-OUTPUT-CHANNEL layers of the U-Net (full resolution: conv1_2, conv9_1, conv9_2 forward; the
// backward-data of conv9_2, conv1_2 and conv2_1), gfx950.  Same contract and arithmetic as conv_bfd_kernel (conv_bfd.hip): bf16 NHWC activations,
// fp32 accumulation on v_mfma_f32_32x32x16_bf16, bias / LeakyReLU / slope / fused 2x2 max-pool epilogues -- models/arch/Unet.py:11-12,44-46,49-51,83-88
// and their autograd backward-data.
//
// These launches are HBM-bound, not MFMA-bound: 32 output channels give 144 flop per activation byte (72 per byte for the 64-channel input of
// conv9_1) against a machine balance of ~400 -- one 16 x 32 pixel tile moves 39 KB in and 32 KB out for 36 MFMAs per wave.  So the kernel is built
// around the activation stream instead of the matrix pipe:
//   * the WHOLE packed weight tensor of the layer (9 taps x 32 x K <= 36 KB, conv_bfd's slab layout at BN = 32, laid out by the pack kernel as the
//     exact LDS image) is copied into LDS once per workgroup and stays there: no weight stream, and the only LDS-DMA traffic of the main loop is
//     the activation halo tile;
//   * halo tiles (18 x 34 pixels x one 32-channel chunk = 39 KB, conv_bfd's XOR-swizzled landing order) run through a ring of THREE buffers: the
//     tiles of work items j+1 and j+2 are in flight while item j is being multiplied -- 78 KB of loads outstanding per CU, what it takes to cover
//     the ~2.7 us DMA round trip at ~25 GB/s per CU.  A work item is (tile, 32-channel chunk); the ring runs across tiles, so the pipeline never
//     drains inside a launch;
//   * one barrier per work item (all nine taps of a chunk: 36 MFMAs per wave) instead of one per kernel row;
//   * each wave waits for its own pieces with a counted s_waitcnt: at the top of item j at most the A_IT pieces of item j+1 may be outstanding.
//     The count is deliberately the conservative one: it does not rely on loads and stores retiring in one common order (the epilogue's stores
//     sit between two tiles' pieces in the queue) -- "at most A_IT operations outstanding" implies "at most A_IT loads outstanding" either way.
#include <stdlib.h>
#include <atomic>
#include "conv.h"

#define TW 32
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {

__device__ __forceinline__ void bfs_dma16(i32x4 rsrc, unsigned voff, unsigned soff, unsigned lds_dst) {      // as conv_bfd.hip::bfd_dma16
    unsigned keep;
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(soff), "s"(lds_dst) : "memory");
}
template <int N>
__device__ __forceinline__ void bfs_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

constexpr int BFS_NAB = 3;                               // activation ring depth
constexpr int BFS_WCHUNK = 9 * 32 * 64;                  // packed weights of one 32-channel chunk: 18 DMA pieces

// ACT: EPI_GRAD with a slope epilogue -- 1: from the saved activation (four 16-byte loads per row), 2: from its 2-bit slope codes (conv.h
// ConvArgs::codes0: one 4-byte load per row) -- compile time, because these loads are hand-issued asm loads whose destination registers must not
// pass through a phi (cdna_hip_programming.md 5.7 item 1: the compiler may copy them before the data has landed)
template <int RPW, int WAVES, int ACT>
__global__ __launch_bounds__(64 * WAVES) void conv_bfs_kernel(const ConvArgs a) {
    constexpr int TH = WAVES * RPW, HW2 = TW + 2, A_PIX = (TH + 2) * HW2;
    constexpr int A_UNITS = A_PIX * 4, A_PIECES = (A_UNITS + 63) / 64, A_BYTES = A_PIECES * 1024;
    constexpr int A_IT = (A_PIECES + WAVES - 1) / WAVES;
    extern __shared__ __attribute__((aligned(16))) char lds[];           // [A0][A1][A2][W chunk 0][W chunk 1]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 31, hi = lane >> 5;
    const int Cin = a.C0 + a.C1, NCH = Cin >> 5, NCH0 = a.C0 >> 5;       // NCH = 1 or 2
    const int tiles_img = a.tiles_x * a.tiles_y;
    const int total_tiles = tiles_img * a.N;
    const int Cs0 = a.C0;

    const unsigned lds_base = (unsigned)__builtin_amdgcn_readfirstlane((int)(size_t)(__attribute__((address_space(3))) char*)lds);
    const unsigned ldsW_addr = lds_base + BFS_NAB * A_BYTES;
    constexpr unsigned OOB = 0xFFFFFFF0u;

    // ---- the layer's weights: 18 * NCH linear 1 KiB pieces, dealt round-robin (a duplicate piece rewrites identical bytes) -------------------
    {
        const unsigned long long wbase = (unsigned long long)a.wp;
        const i32x4 rsrc_w = {(int)(unsigned)wbase, (int)((unsigned)(wbase >> 32) & 0xFFFFu), NCH * BFS_WCHUNK, 0x00020000};
        const int wpieces = 18 * NCH;
#pragma unroll
        for (int it = 0; it < 5; ++it) {                                  // ceil(36 / 8) = 5; with NCH = 1 the later ones repeat earlier pieces
            const int piece = (wave + it * WAVES) % wpieces;
            bfs_dma16(rsrc_w, (unsigned)lane * 16u, (unsigned)(piece * 1024), ldsW_addr + (unsigned)(piece * 1024));
        }
    }

    // ---- activation DMA: which (halo pixel, octet) lands in this lane's slot of piece wave + it*WAVES --------------------------------------
    int a_hy[A_IT], a_hx[A_IT];
    unsigned a_oct[A_IT];
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
        const int piece = (wave + it * WAVES) % A_PIECES;
        const int u = piece * 64 + lane;
        const int P = u >> 2;
        const int hr = P / HW2, hc = P - hr * HW2;
        a_hy[it] = u < A_UNITS ? hr - 1 : -1000;
        a_hx[it] = hc - 1;
        a_oct[it] = (unsigned)((u & 3) ^ ((hc >> 2) & 3)) * 16u;
    }
    unsigned a_voff[A_IT];
    int l_tile = -1, l_img = 0;
    auto decode = [&](int t, int& img, int& y0, int& x0) {
        img = t / tiles_img;
        const int r = t - img * tiles_img;
        int ty, tx;
        band_tile(r, a.tiles_x, a.tiles_y, a.band, ty, tx);
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
    const int n_items = my_tiles * NCH;
    // item j = (tile first + (j / NCH) * stride, chunk j % NCH); its halo tile lives in ring slot j % 3
    // Past the last item the same A_IT instructions are issued with out-of-range offsets (zeros land in a ring slot nobody reads again): every
    // item then has exactly A_IT younger DMA instructions behind it, and every wait of the loop is the same immediate.
    auto issue_A = [&](int j) {
        const int k = NCH == 2 ? (j >> 1) : j, chunk = NCH == 2 ? (j & 1) : 0;
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
        const int buf = j % BFS_NAB;
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            const int piece = (wave + it * WAVES) % A_PIECES;
            bfs_dma16(rsrc_a, a_voff[it], (unsigned)(cs * 64), lds_base + (unsigned)(buf * A_BYTES + piece * 1024));
        }
    };

    // ---- fragment addresses -----------------------------------------------------------------------------------------------------------------
    unsigned fx_off[3][2], fw_off[2];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const int hc = m + kx;
            fx_off[kx][kb] = (unsigned)(hc * 64 + (((kb * 2 + hi) ^ ((hc >> 2) & 3)) * 16));
        }
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) fw_off[kb] = (unsigned)(m * 64 + (((kb * 2 + hi) ^ ((m >> 2) & 3)) * 16));
    const char* ldsW = lds + BFS_NAB * A_BYTES;

    // bias of this lane's 16 channels, loaded once (a load inside the loop's epilogue would drain the DMA queue ahead of it: vmcnt is in order)
    float4 bs[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) bs[q] = a.epi == EPI_FWD ? *reinterpret_cast<const float4*>(a.bias + 4 * hi + 8 * q) : make_float4(0.f, 0.f, 0.f, 0.f);

#pragma unroll
    for (int q = 0; q < 4; ++q) asm volatile("" : "+v"(bs[q].x), "+v"(bs[q].y), "+v"(bs[q].z), "+v"(bs[q].w));      // the loads complete HERE, not at a vmcnt(0) inside the loop
    const float sl = a.lrelu ? 0.2f : 1.0f;                // max(1 v, v) = v
    static_assert(RPW == 2, "the activation-load wait names ac[2][2]");
    u32x4 ac[RPW][2];                                      // native vector type: one 128-bit register tuple per asm operand
    unsigned cw[RPW];                                      // ACT == 2: this lane's code word of each row (16 channels x 2 bits)

    issue_A(0);
    issue_A(1);

    f32x16 acc[RPW];
    for (int j = 0; j < n_items; ++j) {
        const int k = NCH == 2 ? (j >> 1) : j, chunk = NCH == 2 ? (j & 1) : 0;
        const int t = first + k * stride;
        // this wave's pieces of item j (and, the first time, of the weights) have landed; item j+1's may still fly
        bfs_wait_vm<A_IT>();
        __syncthreads();                         // ... and everybody else's; everybody is done reading ring slot (j - 1) % 3 = (j + 2) % 3
        if constexpr (ACT == 2) {                 // slope codes of this tile's output pixels: word (pixel x0 + m, hi) -- the MFMA result layout itself
            int img, y0, x0;
            decode(t, img, y0, x0);
            const int xx = min(x0 + m, a.W - 1);
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                const int y = min(y0 + wave * RPW + r, a.H - 1);
                const unsigned* p = a.codes0 + ((size_t)(img * a.H + y) * a.W + xx) * 2 + hi;
                asm volatile("global_load_dword %0, %1, off" : "=&v"(cw[r]) : "v"(p) : "memory");     // (hand-issued for the same reason as ACT == 1's; early clobber: not the address pair, whose registers the compiler reuses as don't-care operands)
            }
        }
        if constexpr (ACT == 1) {                 // (NCH == 1 in these launches: every item ends a tile)
            // saved activations of this tile's output pixels (slope epilogue), in the line layout; loaded by hand so that the compiler does not
            // wait for them with a vmcnt that would drain the younger halo DMAs: they are OLDER than item j+2's pieces, so the epilogue's
            // wait leaves exactly those A_IT pieces in flight.  Out-of-range pixels re-read a valid address (the value is not used).
            int img, y0, x0;
            decode(t, img, y0, x0);
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                const int y = min(y0 + wave * RPW + r, a.H - 1);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int xx = min(x0 + (lane & 15) + 16 * i, a.W - 1);
                    const bf16_t* p = static_cast<const bf16_t*>(a.act0) + ((size_t)(img * a.H + y) * a.W + xx) * 32 + 8 * bf16_line_group(lane);
                    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(ac[r][i]) : "v"(p) : "memory");
                }
            }
        }
        if (!(ELD_DBG(a) & 4)) issue_A(j + 2);      // (every wave at the top: issuing the SIMD partners' pieces behind the first kernel row, as conv_bfd / conv_bfw do, measured neutral to +2 % here)
        if (chunk == 0) {
#pragma unroll
            for (int r = 0; r < RPW; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[r][i] = 0.f;
        }
        const char* la0 = lds + (j % BFS_NAB) * A_BYTES + (wave * RPW) * (HW2 * 64);
        const char* lw0 = ldsW + (NCH == 2 ? chunk : 0) * 6144;             // slab(ky, chunk) at (ky * NCH + chunk) * 6144
        if (!(ELD_DBG(a) & 2))
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const char* la = la0 + ky * (HW2 * 64);
            const char* lw = lw0 + ky * NCH * 6144;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    uint4 fx[RPW];
#pragma unroll
                    for (int r = 0; r < RPW; ++r) fx[r] = *reinterpret_cast<const uint4*>(la + r * (HW2 * 64) + fx_off[kx][kb]);
                    const uint4 fw = *reinterpret_cast<const uint4*>(lw + kx * (32 * 64) + fw_off[kb]);
#pragma unroll
                    for (int r = 0; r < RPW; ++r)
                        acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fw), __builtin_bit_cast(bf16x8, fx[r]), acc[r], 0, 0, 0);      // D[channel][pixel]
                }
        }
        if constexpr (!ACT) { if (chunk + 1 < NCH) continue; }      // (ACT launches have one chunk per tile: no path from the activation loads past their wait)
        if constexpr (ACT == 2) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(cw[0]), "+v"(cw[1]) : "n"(A_IT));
        if constexpr (ACT == 1) {                // the hand-issued activation loads have landed (item j+2's A_IT DMA pieces, issued after them, may still fly)
            // (whole 128-bit tuples as operands: with sixteen 32-bit operands the compiler shuffled the registers -- v_mov copies of data that had
            // not landed yet -- to build the statement's operand list)
            asm volatile("s_waitcnt vmcnt(%4)" : "+v"(ac[0][0]), "+v"(ac[0][1]), "+v"(ac[1][0]), "+v"(ac[1][1]) : "n"(A_IT));
        }
        if (ELD_DBG(a) & 1) continue;
        // ---- epilogue of tile t.  The MFMA leaves lane (m, hi) with channels 8q + 4hi .. +3 of pixel x0 + m; the stores (and the loads of the saved
        //      activations) use the full-line layout of conv.h bf16_line_swap: instruction i of a wave covers pixels x0 + 16 i .. + 15 completely
        //      (1 KiB contiguous), lane l holding pixel (l & 15) + 16 i, channel group bf16_line_group(l) ---------------------------------------------
        int img, y0, x0;
        decode(t, img, y0, x0);
        const int lp = lane & 15, lg = bf16_line_group(lane);
        if (a.epi == EPI_FWD) {                 // bias + max(0.2 v, v) once, in place (the pooled copy below reuses the activated values)
#pragma unroll
            for (int r = 0; r < RPW; ++r)
#pragma unroll
                for (int q = 0; q < 4; ++q) bias_lrelu4(acc[r], 4 * q, bs[q], sl);
        }
        bool all_nz = true;
        if constexpr (ACT == 2) all_nz = __builtin_amdgcn_ballot_w64((cw[0] & cw[1] & BF16_CODES_NZ_ALL) != BF16_CODES_NZ_ALL) == 0;
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const int y = y0 + wave * RPW + r;
            float4 v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = make_float4(acc[r][4 * q], acc[r][4 * q + 1], acc[r][4 * q + 2], acc[r][4 * q + 3]);
            if constexpr (ACT == 2) {                                       // EPI_GRAD: times the LeakyReLU slope its code names (the same three values)
                float f[16];
                slopes_of_bf16_codes(cw[r], all_nz, f);
#pragma unroll
                for (int q = 0; q < 4; ++q) { v[q].x *= f[4 * q]; v[q].y *= f[4 * q + 1]; v[q].z *= f[4 * q + 2]; v[q].w *= f[4 * q + 3]; }
            }
            if constexpr (ACT == 1) {                                       // EPI_GRAD: times the LeakyReLU slope of the saved activation
                uint2 sp[4];
                bf16_line_unswap(make_uint4(ac[r][0][0], ac[r][0][1], ac[r][0][2], ac[r][0][3]), make_uint4(ac[r][1][0], ac[r][1][1], ac[r][1][2], ac[r][1][3]), sp);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 sv = unpack_bf4(sp[q]);
                    v[q].x *= lrelu_slope(sv.x); v[q].y *= lrelu_slope(sv.y); v[q].z *= lrelu_slope(sv.z); v[q].w *= lrelu_slope(sv.w);
                }
            }
            uint2 pk[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) pk[q] = pack_bf4(v[q]);
            if constexpr (ACT == 0) {
                if (a.codes_out != nullptr && a.epi == EPI_FWD && y < a.H && x0 + m < a.W) {      // slope codes of the ROUNDED activations (wave-uniform pointer test)
                    a.codes_out[((size_t)(img * a.H + y) * a.W + x0 + m) * 2 + hi] = slope_codes_bf16(pk);
                }
            }
            uint4 s0, s1;
            bf16_line_swap(pk, s0, s1);                                     // every lane takes part; only the stores are predicated
            bf16_t* row = static_cast<bf16_t*>(a.out0) + ((size_t)(img * a.H + y) * a.W + x0) * 32 + 8 * lg;
            if (y < a.H && x0 + lp < a.W) *reinterpret_cast<uint4*>(row + lp * 32) = s0;
            if (y < a.H && x0 + lp + 16 < a.W) *reinterpret_cast<uint4*>(row + (lp + 16) * 32) = s1;
        }
        const int x = x0 + m;
        const bool xok = x < a.W;
        // fused nn.MaxPool2d(2) (Unet.py:51): vertical pair in the lane's own rows, horizontal pair in lane ^ 1, on the activated fp32 values
        // (max commutes with the monotone bf16 rounding, so this equals pooling the stored tensor)
        if (a.epi == EPI_FWD && a.pool_out != nullptr && xok) {
            const int Hp = a.H >> 1, Wp = a.W >> 1;
#pragma unroll
            for (int rp = 0; rp < RPW / 2; ++rp) {
                const int y = y0 + wave * RPW + 2 * rp;
                if (y >= a.H) continue;
                uint2 pk[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float u[4];
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) u[jj] = fmax_lane_xor1(fmax_raw(acc[2 * rp][4 * q + jj], acc[2 * rp + 1][4 * q + jj]));
                    pk[q] = pack_bf4(make_float4(u[0], u[1], u[2], u[3]));
                }
                const uint4 w0 = bf16_pair_swap(pk[0], pk[1]), w1 = bf16_pair_swap(pk[2], pk[3]);
                if (!(x & 1)) {
                    bf16_t* dp = static_cast<bf16_t*>(a.pool_out) + ((size_t)(img * Hp + (y >> 1)) * Wp + (x >> 1)) * 32 + 8 * hi;
                    *reinterpret_cast<uint4*>(dp) = w0;
                    *reinterpret_cast<uint4*>(dp + 16) = w1;
                }
            }
        }
    }
}

template <int RPW, int WAVES, int ACT>
int launch_bfs(ConvArgs a, hipStream_t st) {
    a.xcd = eld_xcd_mask() & XCD_BF16;
    a.band = eld_tile_band();
    constexpr int TH = WAVES * RPW;
    a.tiles_x = (a.W + TW - 1) / TW;
    a.tiles_y = (a.H + TH - 1) / TH;
    constexpr size_t A_BYTES = (size_t)(((TH + 2) * (TW + 2) * 4 + 63) / 64) * 1024;
    const int NCH = (a.C0 + a.C1) >> 5;
    const size_t lds_bytes = BFS_NAB * A_BYTES + (size_t)NCH * BFS_WCHUNK;
    const long long tiles = (long long)a.tiles_x * a.tiles_y * a.N;
    if (tiles <= 0) return 0;
    if (tiles > 0x3fffffffLL) return ELD_ENOTSUP;
    auto kern = conv_bfs_kernel<RPW, WAVES, ACT>;
    static EldAttrOnce once;
    { const int rc = once.ensure(kern, BFS_NAB * A_BYTES + 2 * (size_t)BFS_WCHUNK); if (rc) return rc; }
    long long grid = (long long)eld_num_cus();
    if (grid > tiles) grid = tiles;
    ELD_LAUNCH(kern, dim3((unsigned)grid), dim3(64 * WAVES), lds_bytes, st, a);
    ELD_LAUNCH_CHECK();
    return 0;
}

}  // namespace

// Layers this kernel takes: bf16 3x3 with exactly 32 output channels (GEMM N), K = 32 or 64 input channels (one tensor, or the virtual concat of two
// 32-channel tensors), a single output tensor, on a tile domain that gives every CU a tile; weights in conv_bfd's slab layout at BN = 32.
int debug_kernel_mask(int set) {
    static std::atomic<int> mask([] { const char* e = getenv("ELD_DEBUG_KERNEL_MASK"); return e ? atoi(e) : 0; }());      // (env: same-box A/B runs of bench.py)
    return set >= 0 ? mask.exchange(set) : mask.load();
}
extern "C" int eld_debug_kernel_mask(int mask) { return debug_kernel_mask(mask < 0 ? 0 : mask); }

bool bfs_takes(int Nout, int K, int N, int H, int W) {
    if (debug_kernel_mask(-1) & 1) return false;
    if (Nout != 32 || (K != 32 && K != 64)) return false;
    const long long px_tiles = (long long)((W + TW - 1) / TW) * ((H + 15) / 16) * N;
    return px_tiles >= eld_num_cus();
}

int launch_conv_bfs(const ConvArgs& a, hipStream_t st) {
    if ((size_t)a.H * a.W * a.C0 * 2 >= 0xFFFFFFF0ull) return ELD_ENOTSUP;
    if (a.pool_out && (a.epi != EPI_FWD || (a.H & 1) || (a.W & 1))) return ELD_EINVAL;
    if (a.epi == EPI_GRAD && (a.split != 32 || a.out1 != nullptr)) return ELD_ENOTSUP;
    if (a.epi != EPI_FWD && a.epi != EPI_GRAD) return ELD_ENOTSUP;
    if (a.epi == EPI_GRAD && a.act0 != nullptr) {
        if (a.C0 + a.C1 != 32) return ELD_ENOTSUP;      // the slope variant assumes one chunk per tile (the U-Net's two such launches have K = 32)
        return a.codes0 ? launch_bfs<2, 8, 2>(a, st) : launch_bfs<2, 8, 1>(a, st);
    }
    return launch_bfs<2, 8, 0>(a, st);
}
