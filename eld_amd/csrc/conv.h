// conv.h -- argument records and launchers of the U-Net convolution kernels (internal; gfx950).
//
// Activations are NHWC float32 in HBM (channel-contiguous: the GEMM K dimension of the implicit
// GEMM is contiguous, a 32-channel pixel is one 128-byte line).  All three convolution flavours of
// the U-Net (3x3 s1 p1, 1x1, and the 2x2 stride-2 transposed conv in both directions) are one
// implicit-GEMM kernel family on the exact-fp32 MFMA v_mfma_f32_32x32x2_f32:
//      D[pixel][n] = sum_{tap, c} A_tap[pixel][c] * Wp[tap][n][c]
#pragma once
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_hw __attribute__((ext_vector_type(2)));

enum ConvMode { CONV_3X3 = 0, CONV_1X1 = 1, CONV_GATHER2X2 = 2 };
enum ConvEpi { EPI_FWD = 0, EPI_CONVT_FWD = 1, EPI_GRAD = 2 };

typedef unsigned short bf16_t;       // storage type of a bfloat16 element
enum ConvDtype { DT_F32 = 0, DT_BF16 = 1 };

// Element type T of activations / packed weights is float or bf16_t (dtype); accumulation, bias are always float.
struct ConvArgs {
    const void* in0;    // NHWC source 0 (C0 channels)
    const void* in1;    // NHWC source 1 (C1 channels) -- virtual channel concat [in0, in1]; may be null
    int C0, C1;
    const void* wp;     // packed weights [taps][Nout][C0+C1], c contiguous
    int N, H, W;        // tile domain: output pixels (3x3 / 1x1) or input-resolution pixels (gather: source is 2H x 2W)
    int Nout;           // GEMM N
    int epi;
    const float* bias;  // EPI_FWD: [Nout]; EPI_CONVT_FWD: [Cout_t]
    int lrelu;          // EPI_FWD: apply max(0.2v, v)
    void* out0;         // EPI_FWD/CONVT: destination.  EPI_GRAD: channels [0, split)
    void* out1;         // EPI_GRAD: channels [split, Nout)
    int band;           // > 1: tile ids run in bands of `band` tile rows, column-major inside a band (band_tile below)
    int xcd;            // != 0: workgroup ids are remapped so that every XCD takes a CONTIGUOUS range of the launch's work items (xcd_block below)
    int ksplit;         // conv_x3d_kernel, small problems: > 1 = split the K (input-channel chunk) range of every tile over this many workgroups; the
    float* kpart;       //   partial sums go to kpart[ksplit][N][H][W][Nout] (fp32) and x3_splitk_finish_kernel adds them in a fixed order and runs the
    size_t kpart_floats;//   epilogue.  kpart_floats = capacity of kpart (0: no split).  Set by the U-Net orchestration only.
    void* pool_out;     // EPI_FWD, optional: also write the 2x2/stride-2 max-pool of the output ([N, H/2, W/2, Nout]; H, W even) -- honoured by the
                        // three-piece 3x3 kernels (conv_x3.hip) only; other launchers return ELD_ENOTSUP when it is set
    int split;
    const void* act0;   // EPI_GRAD: saved post-activation tensor (layout of out0) -> multiply by lrelu slope; may be null
    const void* act1;
    // Slope codes (round 5, fp32 three-piece kernels): 2 bits per element of an activation tensor -- bit 0 = negative, bit 1 = zero -- 16 elements per 32-bit word
    // in the MFMA result layout: word ((pixel * (C / 32) + block) * 2 + hi) holds, at bits 2i .. 2i+1 (i = 4q + j), channel 8q + 4hi + j of the 32-channel block.
    // EPI_FWD with lrelu: codes_out != null -> also written for the layer's output.  EPI_GRAD: codes0 / codes1 != null -> read INSTEAD of act0 / act1
    // (1/16 of the bytes, one load per row and block instead of four).
    // The bf16 network (conv_bfs.hip, conv_first.hip) keeps the same word addressing for its two 32-channel tensors but another bit order inside a word
    // (slope_codes_bf16 below).
    unsigned* codes_out;
    // Pool codes (round 6): with pool_out, also the ARGMAX of every 2x2 window -- what the pool's backward needs of the un-pooled tensor besides its slope codes.
    // Word ((pooled pixel * (Nout / 32) + block) * 2 + hi): bit i (i = 4q + j <-> channel 8q + 4hi + j of the block) = the winner lies in the window's BOTTOM row,
    // bit 16 + i = in its RIGHT column; the winner is the first maximum in row-major window order (torch's max_pool2d backward; conv_x3_dev.h pool_epilogue).
    unsigned* pool_codes_out;
    const unsigned* codes0;
    const unsigned* codes1;
    int Cout_t;         // EPI_CONVT_FWD: real Cout (Nout = 4*Cout_t, n = tap*Cout_t + co)
    int tiles_x, tiles_y;
    int vp;             // virtual-row pitch of the strip tiles_y was counted on (vrow_pitch; set by the launchers that tile the strip)
    int tile_h, tile_w; // conv_x3d / conv_bfd: the launch's tile shape (rows x columns of pixels; tile_h * tile_w <= the kernel's pixel count)
    int dtype;          // DT_F32 / DT_BF16
    int algo;           // fp32 product scheme of THIS call: 0 fp32 MFMA, 1 three bf16 pieces, 2 two fp16 pieces; < 0 = process default (conv_fp32_algo)
    int dbg;            // ablation switches for tools/ (env ELD_CONV_DBG): 1 skip epilogue, 4 skip staging loads, 8/16 skip slab/halo stores (conv_x3), 64 one workgroup per CU
    unsigned long long* prof;   // dev tool (eld_debug_conv_prof): per-stage s_memtime stamps of the first workgroups; null in production
    // two-piece fp16 product scheme (conv_fp32_algo 2): device floats holding an upper bound of max|.| of each operand tensor
    // (read as power-of-two scales that bring the operands into the fp16 range) and, optionally, where to accumulate
    // max|out| of what this launch writes (atomic max on the bit pattern; the slot is zeroed by the caller)
    const float* amax_in0; const float* amax_in1; const float* amax_w;
    float* amax_out0; float* amax_out1;
};

// d/dx max(0.2x, x) as autograd computes it for torch.max(0.2*x, x) (models/arch/Unet.py:102-104):
// ties (x == 0) split the gradient evenly between the two branches -> 0.6.  The saved tensor is the
// post-activation value, whose sign equals the pre-activation's.
// power-of-two scale that maps a tensor with max|x| <= *slot into [2^14, 2^15) (fp16 holds 65504); 1 for an all-zero tensor
__device__ __forceinline__ float h2_scale(float amax) {
    const int e = (int)((__float_as_uint(amax) >> 23) & 0xFFu);
    if (e == 0) return 1.0f;
    int se = 268 - e;                                   // biased exponent of 2^(14 - (e - 127))
    se = se < 1 ? 1 : (se > 254 ? 254 : se);
    return __uint_as_float((unsigned)se << 23);
}
// atomic max of |v| over a wave into *slot (non-negative floats order like their bit patterns)
__device__ __forceinline__ void amax_accumulate(float* slot, float v) {
    unsigned b = __float_as_uint(v) & 0x7FFFFFFFu;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const unsigned o = __shfl_xor(b, off, 64); b = b > o ? b : o; }
    // the plain read first keeps almost every wave off the atomic (the slot only grows; a stale read just costs one atomic)
    if ((threadIdx.x & 63) == 0 && b > *reinterpret_cast<volatile unsigned*>(slot)) atomicMax(reinterpret_cast<unsigned*>(slot), b);
}

// v_max_f32 without the canonicalising v_max_f32 x, x, x that fmaxf() gets in front of it for every operand the compiler cannot prove quiet
// (results of packed adds, DPP moves, bit casts): operands here are finite
__device__ __forceinline__ float fmax_raw(float a, float b) {
    asm("v_max_f32 %0, %0, %1" : "+v"(a) : "v"(b));
    return a;
}
// max with lane ^ 1 (the horizontal neighbour pixel of the fused 2x2 max-pool): quad_perm [1,0,3,2] folded into the v_max (the builtin pair
// compiles to v_mov_dpp + canonicalise + v_max).  s_nop 1: a VALU write needs two wait states before a DPP read of the same register.
__device__ __forceinline__ float fmax_lane_xor1(float f) {
    float r;
    asm("s_nop 1\n\tv_max_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(f));
    return r;
}
// in place on elements i .. i+3 of an MFMA accumulator: v <- max(sl * t, t), t = v + b: bias and LeakyReLU (sl = 0.2; sl = 1 leaves t) with
// the packed fp32 add / multiply
__device__ __forceinline__ void bias_lrelu4(f32x16& acc, int i, float4 b, float sl) {
    f32x2 t01 = {acc[i], acc[i + 1]}, t23 = {acc[i + 2], acc[i + 3]};
    const f32x2 b01 = {b.x, b.y}, b23 = {b.z, b.w};
    t01 += b01; t23 += b23;
    const f32x2 s01 = t01 * sl, s23 = t23 * sl;
    acc[i] = fmax_raw(t01.x, s01.x); acc[i + 1] = fmax_raw(t01.y, s01.y); acc[i + 2] = fmax_raw(t23.x, s23.x); acc[i + 3] = fmax_raw(t23.y, s23.y);
}
// in place on elements i .. i+3 of an MFMA accumulator: v <- max(sl * v, v) (the bias already sits in the accumulator: it was its initial value)
__device__ __forceinline__ void lrelu4(f32x16& acc, int i, float sl) {
    const f32x2 t01 = {acc[i], acc[i + 1]}, t23 = {acc[i + 2], acc[i + 3]};
    const f32x2 s01 = t01 * sl, s23 = t23 * sl;
    acc[i] = fmax_raw(t01.x, s01.x); acc[i + 1] = fmax_raw(t01.y, s01.y); acc[i + 2] = fmax_raw(t23.x, s23.x); acc[i + 3] = fmax_raw(t23.y, s23.y);
}
__device__ __forceinline__ float lrelu_slope(float y) { return y > 0.f ? 1.0f : (y < 0.f ? 0.2f : 0.6f); }
// 2-bit slope code of a post-activation value (its sign is the pre-activation's): bit 0 = sign bit, bit 1 = zero (+0 or -0); lrelu_slope of the code
__device__ __forceinline__ unsigned slope_code(float v) {
    const unsigned b = __float_as_uint(v);
    return (b >> 31) | ((b << 1) == 0u ? 2u : 0u);
}
__device__ __forceinline__ unsigned slope_codes16(const f32x16& acc) {
    unsigned w = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) w |= slope_code(acc[i]) << (2 * i);
    return w;
}
__device__ __forceinline__ float slope_of_code(unsigned w, int i) {
    const unsigned c = w >> (2 * i);
    return (c & 2u) ? 0.6f : ((c & 1u) ? 0.2f : 1.0f);
}

// bfloat16 <-> float.  Rounding to nearest even is the hardware's (v_cvt_pk_bf16_f32, one instruction per PAIR of values): a software
// round costs five VALU instructions per value, which made the bf16 conv epilogues VALU-bound (DESIGN.md, bf16 path).
__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float((unsigned)v << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
__device__ __forceinline__ unsigned pack_bf2(float lo, float hi) {
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_hw));
}
__device__ __forceinline__ uint2 pack_bf4(float4 v) { return make_uint2(pack_bf2(v.x, v.y), pack_bf2(v.z, v.w)); }
// Slope codes of the bf16 network: 16 packed bf16 activations of a lane (pack_bf4 of accumulator elements 4q .. 4q+3 -> pk[q]; element e = 4q + b) ->
// one 32-bit word, 2 bits per element, taken from the ROUNDED values (the ones the bf16 backward's lrelu_slope(saved activation) would see: a tiny
// fp32 value may round to a bf16 zero).  The bit order is whatever is cheapest to build from the packed words -- the epilogues that write these words
// sit on the critical path of HBM-latency-bound kernels (conv_bfs.hip), and the element-by-element form of slope_codes16 cost 11 % of conv9_1:
//   sign of element e      at bit 8 (e & 3) + (e >> 2)                        (v_perm_b32 gathers the four high bytes of pk[q]; one mask and shift per q)
//   "nonzero" of element e at bit P(e >> 1) + 16 (e & 1), P(k) = k + 4 + 4 (k >> 2)   (v_pk_min_u16(|half|, 1) leaves the flags of a word's halves at bits 0 / 16)
// i.e. signs in the low nibble of every byte, nonzero flags in the high nibbles.  slope = nonzero ? (sign ? 0.2 : 1) : 0.6 (conv.h lrelu_slope; -0 is a zero).
constexpr unsigned BF16_CODES_NZ_ALL = 0xF0F0F0F0u;
__device__ __forceinline__ unsigned slope_codes_bf16(const uint2 (&pk)[4]) {
    unsigned sg = 0, nz = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const unsigned hb = __builtin_amdgcn_perm(pk[q].y, pk[q].x, 0x07050301u);      // high bytes of elements 4q .. 4q+3
        sg |= (hb & 0x80808080u) >> (7 - q);
        const unsigned w[2] = {pk[q].x, pk[q].y};
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int k = 2 * q + h;                                                    // word k holds elements 2k, 2k + 1
            unsigned f;                                                                  // (hipcc expands a two-lane umin into compares, selects and a perm: name the instruction)
            asm("v_pk_min_u16 %0, %1, %2" : "=v"(f) : "v"(w[h] & 0x7fff7fffu), "v"(0x00010001u));
            nz |= f << (k + 4 + 4 * (k >> 2));
        }
    }
    return sg | nz;
}
// the slopes of a lane's 16 elements from its code word: f[e] = sign ? 0.2 : 1 as a bit-field extract and a bit select (two instructions); all_nz
// (wave-uniform: no element of this wave's words is a zero -- the usual case) skips the 0.6 fix-up of exact zeros
__device__ __forceinline__ void slopes_of_bf16_codes(unsigned w, bool all_nz, float (&f)[16]) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int sp = 8 * (e & 3) + (e >> 2);
        unsigned neg;                                                                   // all ones for a negative value (named instruction: written as shifts,
        asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(neg) : "v"(w), "n"(sp));                    // hipcc turns the whole expression back into and + compare + select)
        asm("v_bfi_b32 %0, %1, %2, 1.0" : "=v"(f[e]) : "v"(neg), "s"(0x3E4CCCCDu));       // neg ? 0.2f : 1.0f
    }
    if (!all_nz) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int np = (e >> 1) + 4 + 4 * (e >> 3) + 16 * (e & 1);
            f[e] = ((w >> np) & 1u) ? f[e] : 0.6f;
        }
    }
}
__device__ __forceinline__ float4 unpack_bf4(uint2 p) {
    return make_float4(__uint_as_float(p.x << 16), __uint_as_float(p.x & 0xFFFF0000u), __uint_as_float(p.y << 16), __uint_as_float(p.y & 0xFFFF0000u));
}

// Widened bf16 epilogue store (cdna_hip_programming.md T21).  After a 32x32 MFMA run as D[channel][pixel], lane (m, hi) holds, for each
// 8-channel group q of a 32-channel block, channels 8q + 4hi .. +3 of pixel m: packed, that is 8 bytes per lane and group -- a row-per-lane
// dwordx2 store, which is store-ISSUE bound (~7 B/clk/CU).  One v_permlane32_swap per dword exchanges the halves of a group PAIR (2j, 2j+1):
// afterwards lanes hi = 0 hold all 8 channels of group 2j and lanes hi = 1 all 8 of group 2j+1 -> one 16-byte store per lane and pair, at
// channel offset 8 * (2j + hi) of the block.  a = this lane's packed group 2j, b = its packed group 2j+1.  Both lanes of a pair (m, m+32) must
// be active (they are the same pixel: bounds tests depend on m only).
__device__ __forceinline__ uint4 bf16_pair_swap(uint2 a, uint2 b) {
    const auto r0 = __builtin_amdgcn_permlane32_swap(a.x, b.x, false, false);
    const auto r1 = __builtin_amdgcn_permlane32_swap(a.y, b.y, false, false);
    return make_uint4(r0[0], r1[0], r0[1], r1[1]);
}

// Full-line variant for a row of 32 pixels x one 32-channel block.  In: pk[q] = this lane's packed channels 8q + 4hi .. +3 of pixel m = lane & 31.
// Out: two 16-byte values; value i belongs to pixel (lane & 15) + 16 i of the row, channel group bf16_line_group(lane) (8 channels, 16 bytes).
// Store instruction i of a wave then covers 16 CONSECUTIVE pixels x 64 bytes -- 1 KiB of contiguous memory when the tensor has 32 channels --
// instead of 32 bytes of each of 32 pixels.  Built from bf16_pair_swap plus one v_permlane16_swap per dword (rows of 16 lanes: row 1 of the first
// operand <-> row 0 of the second, row 3 <-> row 2).  bf16_line_unswap is the inverse (both networks are involutions): the same two 16-byte values,
// loaded in the line layout, back to the lane's own quarter-groups -- used for the saved activations of the slope epilogue.
__device__ __forceinline__ int bf16_line_group(int lane) { return ((lane >> 4) & 1) * 2 + (lane >> 5); }      // rows 0..3 of 16 lanes hold groups 0, 2, 1, 3
__device__ __forceinline__ void bf16_rows_swap(uint4& g0, uint4& g1) {
    auto r = __builtin_amdgcn_permlane16_swap(g0.x, g1.x, false, false); g0.x = r[0]; g1.x = r[1];
    r = __builtin_amdgcn_permlane16_swap(g0.y, g1.y, false, false); g0.y = r[0]; g1.y = r[1];
    r = __builtin_amdgcn_permlane16_swap(g0.z, g1.z, false, false); g0.z = r[0]; g1.z = r[1];
    r = __builtin_amdgcn_permlane16_swap(g0.w, g1.w, false, false); g0.w = r[0]; g1.w = r[1];
}
__device__ __forceinline__ void bf16_line_swap(const uint2 (&pk)[4], uint4& s0, uint4& s1) {
    s0 = bf16_pair_swap(pk[0], pk[1]);      // lane (m, hi): group hi
    s1 = bf16_pair_swap(pk[2], pk[3]);      // lane (m, hi): group 2 + hi
    bf16_rows_swap(s0, s1);                 // s0: pixels 0..15, s1: pixels 16..31; row r of 16 lanes: group {0, 2, 1, 3}[r]
}
__device__ __forceinline__ void bf16_line_unswap(uint4 s0, uint4 s1, uint2 (&pk)[4]) {
    bf16_rows_swap(s0, s1);                 // back to: s0 = group hi, s1 = group 2 + hi of pixel m (8 channels each)
    auto r = __builtin_amdgcn_permlane32_swap(s0.x, s0.z, false, false); pk[0].x = r[0]; pk[1].x = r[1];
    r = __builtin_amdgcn_permlane32_swap(s0.y, s0.w, false, false); pk[0].y = r[0]; pk[1].y = r[1];
    r = __builtin_amdgcn_permlane32_swap(s1.x, s1.z, false, false); pk[2].x = r[0]; pk[3].x = r[1];
    r = __builtin_amdgcn_permlane32_swap(s1.y, s1.w, false, false); pk[2].y = r[0]; pk[3].y = r[1];
}

// s_waitcnt vmcnt(n) for a wave-uniform RUNTIME n: "at most n of this wave's vector-memory operations outstanding".  The LDS-DMA kernels count
// what may stay in flight behind the operation they need -- the DMA pieces of later stages AND the store instructions their epilogues issued
// (loads and stores retire in one common order, so the stores issued after a piece are simply younger entries of the same queue).  Any value
// below the true count only waits longer; counts beyond the table fall back to 0.
template <int N>
__device__ __forceinline__ void eld_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void eld_wait_vmcnt_dyn(int n) {
    switch (n) {
#define ELD_W(k) case k: eld_wait_vmcnt<k>(); break;
        ELD_W(1) ELD_W(2) ELD_W(3) ELD_W(4) ELD_W(5) ELD_W(6) ELD_W(7) ELD_W(8) ELD_W(9) ELD_W(10) ELD_W(11) ELD_W(12) ELD_W(13) ELD_W(14) ELD_W(15) ELD_W(16)
        ELD_W(17) ELD_W(18) ELD_W(19) ELD_W(20) ELD_W(21) ELD_W(22) ELD_W(23) ELD_W(24) ELD_W(25) ELD_W(26) ELD_W(27) ELD_W(28) ELD_W(29) ELD_W(30) ELD_W(31) ELD_W(32)
        ELD_W(33) ELD_W(34) ELD_W(35) ELD_W(36) ELD_W(37) ELD_W(38) ELD_W(39) ELD_W(40) ELD_W(41) ELD_W(42) ELD_W(43) ELD_W(44) ELD_W(45) ELD_W(46) ELD_W(47) ELD_W(48)
#undef ELD_W
        default: eld_wait_vmcnt<0>(); break;
    }
}

// fp32 counterpart of the full-line store.  Lane (m, hi) holds, for q = 0..3, channels 8q + 4hi .. +3 of pixel m of a 32-channel block: float4 v[q] is the
// 16-byte piece 2q + hi of the pixel's 128-byte record.  One v_permlane16_swap per dword re-deals (v[0], v[1]) and (v[2], v[3]) so that a store instruction of
// the wave covers 16 consecutive pixels x 64 contiguous bytes (whole 64-byte sectors) instead of 16 bytes of each of 64 half-pixels: afterwards
//   v[0] / v[2] = pieces {0,2,1,3}[lane >> 4] (+ 4 for v[2]) of pixel  lane & 15,       v[1] / v[3] = the same pieces of pixel (lane & 15) + 16.
// f32_line_store does the exchange (EVERY lane must take part) and the predicated stores.  blk = address of channel 0 of the block at the row's pixel
// x0; pstride = elements between consecutive pixels of the row in the destination; valid_px = number of in-range pixels of the row from x0 on.
__device__ __forceinline__ void f32_rows_swap(float4& a, float4& b) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a.x), __float_as_uint(b.x), false, false); a.x = __uint_as_float(r[0]); b.x = __uint_as_float(r[1]);
    r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a.y), __float_as_uint(b.y), false, false); a.y = __uint_as_float(r[0]); b.y = __uint_as_float(r[1]);
    r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a.z), __float_as_uint(b.z), false, false); a.z = __uint_as_float(r[0]); b.z = __uint_as_float(r[1]);
    r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a.w), __float_as_uint(b.w), false, false); a.w = __uint_as_float(r[0]); b.w = __uint_as_float(r[1]);
}
// 16-byte global store, optionally write-through (sc1: the line leaves the XCD's L2 instead of staying there -- MI355X_MICROARCH.md, stores of each flavour)
__device__ __forceinline__ void st16(float* p, float4 v, bool wt = false) {
    typedef float f4v __attribute__((ext_vector_type(4)));
    const f4v q = {v.x, v.y, v.z, v.w};
    if (wt) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(q) : "memory");      // (dev probe only: measured within noise, profiles/r06_ab_notes.md)
    else *reinterpret_cast<float4*>(p) = v;
}
__device__ __forceinline__ void f32_line_store(float4 (&v)[4], float* blk, size_t pstride, int lane, bool row_ok, int valid_px, bool wt = false) {
    f32_rows_swap(v[0], v[1]);
    f32_rows_swap(v[2], v[3]);
    const int lp = lane & 15;
    float* p0 = blk + (size_t)lp * pstride + 4 * bf16_line_group(lane);      // same row -> piece map {0,2,1,3}
    if (row_ok && lp < valid_px) {
        st16(p0, v[0], wt);
        st16(p0 + 16, v[2], wt);
    }
    if (row_ok && lp + 16 < valid_px) {
        float* p1 = p0 + 16 * pstride;
        st16(p1, v[1], wt);
        st16(p1 + 16, v[3], wt);
    }
}

// ---- virtual rows (round 4): no MFMA work on rows that do not exist --------------------------------------------------------------------
// The N images of a launch are stacked into ONE strip: row y of image i is virtual row i * P + y with pitch P = H + S, where the S = 2 - (H & 1)
// rows between two images hold nothing (they read as the convolution's zero padding and are never stored).  Row tiles are cut from the strip, not
// from each image, so a level of 89 rows costs 8 x 90 / 16 = 45 sixteen-row tiles per batch of eight instead of 8 x 6 = 48 (7.9 % -> 1.1 % of
// padding rows; 178 rows: 7.9 % -> 1.1 %).  P is even, so a row pair (2k, 2k + 1) of the strip is a row pair of one image: the fused 2x2 max-pool
// still finds its vertical neighbour in the lane's own registers.  A tile touches at most two images (TH + 2 <= P): kernels address a window of two
// images through one buffer descriptor (base = the image of the tile's first row), which is why the launchers want two images within 4 GB.
// The pitch is a per-launch choice (vrow_pitch): H + S (tiles straddle image seams) when that needs fewer TH-row tiles than the plain
// per-image tiling, else H rounded up to a multiple of TH (tiles never straddle: exactly the per-image tiling, no separator needed).  Kernels
// tell the two apart by P % TH: when it is 0 every strip row beyond the pitch of the tile's first image is treated as absent (it could only be
// the halo of a tile's last row, which must read zeros or feeds a separator row).
__host__ __device__ inline int vrow_extent(int N, int H, int P) { return (N - 1) * P + H; }      // the last image needs no separator
__host__ __device__ inline int vrow_pitch(int N, int H, int TH) {
    const int flat = H + 2 - (H & 1), own = (H + TH - 1) / TH * TH;
    if (flat < TH + 2) return own;                                       // a tile would span more than two images
    return (vrow_extent(N, H, flat) + TH - 1) / TH < (vrow_extent(N, H, own) + TH - 1) / TH ? flat : own;
}

// Tile shape (rows x columns) of conv_x3d_kernel / conv_bfd_kernel for a launch over N images of H x W, among th * tw <= TH * 32 pixel slots and
// (th + 2)(tw + 2) <= (TH + 2) * 34 halo pixels (what the kernels' LDS holds): the one with the lowest cost = tiles x (1 + 0.01 (c - 1)), c = the LDS
// cycles of an activation-fragment read relative to a conflict-free one.  A ds_read_b128 is serviced in the lane groups {0-3,12-15,20-27} and
// {4-11,16-19,28-31} (+32), and the 16 halo pixels of a group fall on distinct banks only if their linear indices are distinct modulo 16: true
// for 16 consecutive pixels of a row, false where a group's pixels wrap to the next tile row (the halo row is 2 pixels longer than the tile
// row).  So non-standard shapes number their slots group by group (conv_slot_of_lane: lanes of group 0 -> slots 0..15, group 1 -> 16..31; widths
// that are multiples of 16 are then conflict-free: SQ_LDS_BANK_CONFLICT / SQ_ACTIVE_INST_LDS of conv_x3d_kernel<128> fell from 0.90 to 0.06) and the
// cost charges what is left only as a tie-break: with the grouped numbering the fewest-tiles choice and a conflict-weighted one measured the same
// time (profiles/r04_ab_notes.md), so the executed MFMA work decides.  `pooled` launches (fused 2x2 max-pool) keep TH x 32.  conv_tile_count = tiles of that shape.
void conv_tile_shape(int N, int H, int W, int TH, bool pooled, int& th, int& tw);
// pixel slot (0..31 of its MFMA column) of lane m = lane & 31 under the grouped numbering / the identity of the standard TH x 32 tiles
__host__ __device__ inline int conv_slot_of_lane(int m, bool grouped) {
    if (!grouped) return m;
    const int q = m >> 2;                       // quads 0..7 -> slot quads {0, 4, 5, 1, 6, 2, 3, 7}
    const int sq = q == 0 ? 0 : q == 1 ? 4 : q == 2 ? 5 : q == 3 ? 1 : q == 4 ? 6 : q == 5 ? 2 : q == 6 ? 3 : 7;
    return sq * 4 + (m & 3);
}
__host__ __device__ inline bool conv_slots_grouped(int th, int tw, int TH) { return !(th == TH && tw == 32); }
long long conv_tile_count(int N, int H, int W, int TH, bool pooled);

// f32_line_store for pixel slots whose pixels are per-lane: p0 / p1 = channel 0 of the 32-channel block in the pixels of slot lane & 15 and slot
// (lane & 15) + 16 of the MFMA column (nullptr: nothing to store).  EVERY lane must take part in the exchange.
__device__ __forceinline__ void f32_line_store2(float4 (&v)[4], float* p0, float* p1, int lane, bool wt = false) {
    f32_rows_swap(v[0], v[1]);
    f32_rows_swap(v[2], v[3]);
    const int g = 4 * bf16_line_group(lane);      // same row -> piece map {0,2,1,3}
    if (p0 != nullptr) {
        st16(p0 + g, v[0], wt);
        st16(p0 + g + 16, v[2], wt);
    }
    if (p1 != nullptr) {
        st16(p1 + g, v[1], wt);
        st16(p1 + g + 16, v[3], wt);
    }
}

int launch_conv(const ConvArgs& a, int mode, hipStream_t st);
// fp32 3x3 convolutions: 0 = exact-fp32 MFMA (conv_igemm.hip), 1 = three-piece bf16 split on the bf16 MFMA (conv_x3.hip).
// set < 0 only queries.  Initial value from env ELD_FP32_CONV (mfma | x3).  Returns the value in force before the call.
// This is only the DEFAULT for calls that do not name a scheme (ConvArgs::algo < 0); every launcher resolves it once per call.
int conv_fp32_algo(int set);
inline int resolve_algo(int algo) { return (algo >= 0 && algo <= 2) ? algo : conv_fp32_algo(-1); }
int launch_conv_x3(const ConvArgs& a, hipStream_t st);
// three-piece scheme: 3x3 layers whose GEMM N (Nout) is a multiple of 64 take their weights PRE-SPLIT in the slab layout of
// conv_x3d_kernel (conv_x3.hip); returns the slab's channel-block width BN (64 or 128), or 0 for layers that keep fp32 packed weights
int x3_slab_bn(int Nout, int N, int H, int W, int* waves = nullptr);      // (N, H, W): the launch's tile domain -- small problems take smaller tiles
// bytes of one packed layer in slab layout: 9 taps x Nout x K/16 rows of 112 B
__host__ __device__ inline size_t x3_slab_stride(int BN) { return (size_t)(3 * BN * 112 + 1023) / 1024 * 1024; }      // bytes of one (ky, chunk, channel-block) slab
void conv_x3_set_prof(unsigned long long* buf);      // dev tool: 8 workgroups x 4 waves x 128 stages x 6 stamps
int launch_conv_x3_gemm(const ConvArgs& a, int mode, hipStream_t st);
// conv_x3w.hip (round 6): the 32-output-channel full-resolution layers on a workgroup of specialised waves (4 MFMA + 4 load / cut waves); weights in the
// BN = 32 slab layout.  x3w_enabled(): the process-wide switch (ELD_X3W, default on) -- it decides the pack layout of every 32-output-channel layer.
bool x3w_enabled();
bool x3w_takes(const ConvArgs& a);
int launch_conv_x3w(const ConvArgs& a, hipStream_t st);
// bf16 3x3 layers with Nout % 64 == 0 and K % 32 == 0 run on conv_bfd_kernel (conv_bfd.hip: both operands by LDS-DMA) and take their weights in its
// slab layout; returns the slab's channel-block width BN (64 / 128) or 0 for layers that stay on conv_igemm_kernel<bf16_t>
int bfd_slab_bn(int Nout, int K, int N, int H, int W);
__host__ __device__ inline size_t bfd_slab_bytes(int BN) { return (size_t)3 * BN * 64; }
int launch_conv_bfd(const ConvArgs& a, hipStream_t st);      // transposed-conv directions; ELD_ENOTSUP if not covered
// bf16 3x3 layers with exactly 32 output channels and K = 32 / 64 (conv_bfs.hip: weights resident in LDS, three-deep activation ring); they take
// their weights in conv_bfd's slab layout at BN = 32 (bfd_slab_bn returns 32 for them)
int debug_kernel_mask(int set);      // eld_debug_kernel_mask: set < 0 only queries
bool bfs_takes(int Nout, int K, int N, int H, int W);
int launch_conv_bfs(const ConvArgs& a, hipStream_t st);
// bf16 3x3 layers with exactly 64 output channels and K = 32 / 64 (conv_bfw.hip: weights resident in LDS, four-deep ring of 16-channel halo tiles);
// same BN = 64 slab layout as conv_bfd_kernel<64>, chosen per launch inside launch_conv_bfd
bool bfw_takes(const ConvArgs& a);
int launch_conv_bfw(const ConvArgs& a, hipStream_t st);
// bf16 transposed convolutions (conv_bfg.hip: both operands by LDS-DMA): column-block width of the packed slabs (bfg_store), 0 = stays on conv_igemm_kernel
int bfg_slab_bn(bool gather, int Nout, int Cs, int Cout_t, int N, int H, int W);
__host__ __device__ inline size_t bfg_slab_bytes(int BN) { return (size_t)BN * 64; }
int launch_conv_bfg(const ConvArgs& a, int mode, hipStream_t st);

// XCD-aware work-item ids (round 6).  The dispatcher places block b on XCD b % 8 (observed, not promised: MI355X_MICROARCH.md, workgroup dispatch), and every XCD
// has its own 4 MiB L2.  The kernels number their work items so that NEIGHBOURS share operands (the column blocks of one pixel tile, the (out, in) channel
// blocks of one pixel slice of a weight gradient): with the identity map those neighbours sit on eight different XCDs and every one of the eight L2s fetches the
// shared tile from the fabric.  xcd_block() hands XCD x the x-th CONTIGUOUS eighth of the ids instead (bijective for any grid size) -- a pure speed choice:
// nothing depends on where a block runs.  eld_xcd_mask(): env ELD_XCD, one bit per kernel family (see launchers), for same-box A/B runs.
#ifdef __HIPCC__
__device__ __forceinline__ int xcd_block(int on) {
    const int b = (int)blockIdx.x;
    if (!on) return b;
    const int n = (int)gridDim.x, q = n >> 3, r = n & 7, x = b & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
}
#endif
// Tile id -> (tile row, tile column) in BANDS of BH tile rows, column-major inside a band: 32 consecutive ids (what an XCD takes per round under xcd_block) are then an
// 8-column x 4-row block of tiles instead of half a tile row, and the halo rows / columns neighbouring tiles share are fetched once per XCD (the level-0 weight
// gradient reads 6 x 34 pixels of X per 4 x 32 pixels of G).  BH <= 1: row-major.  Bijective for any tiles_x x tiles_y (the last band is shorter).
__host__ __device__ inline void band_tile(int id, int tiles_x, int tiles_y, int BH, int& ty, int& tx) {
    if (BH <= 1) { ty = id / tiles_x; tx = id - ty * tiles_x; return; }
    const int full = tiles_x * BH, band = id / full, off = id - band * full;
    int bh = tiles_y - band * BH;
    if (bh > BH) bh = BH;
    tx = off / bh;
    ty = band * BH + (off - tx * bh);
}
int eld_tile_band();      // env ELD_TILE_BAND (default 4; 1 = row-major)
enum { XCD_GEMM = 1, XCD_WGRAD8 = 2, XCD_WGRAD = 4, XCD_X3D = 8, XCD_X3W = 16, XCD_BF16 = 32, XCD_IGEMM = 64 };
int eld_xcd_mask();

// dW-type reduction:  P[tap][i][j] = sum_pixels G[pixel][i] * X[pixel (+) tap][j]
struct WgradArgs {
    const void* g;       // NHWC, CA channels, unshifted (the A operand); element type = dtype
    int CA;
    const void* x0;      // NHWC sources of the shifted/gathered operand (virtual concat)
    const void* x1;
    int C0, C1;          // C0 + C1 = CB (may be smaller than the padded j extent)
    int N, H, W;         // pixel domain of g
    float* part;         // partials [psplit][taps][CA][CBp]
    float* bpart;        // optional bias partials [psplit][CA] (sum over pixels of g); null to skip
    float* xbpart;       // CONV_GATHER2X2, optional: partials [psplit][CBp] of the column sums of the gathered operand (every pixel
                         // of x0 is staged exactly once per block column) = the transposed conv's bias gradient; null to skip
    int CBp;             // padded CB (multiple of 32)
    int psplit;
    int tiles_x, tiles_y;
    int vp;              // wgrad8_kernel: virtual-row pitch of the strip tiles_y was counted on (vrow_pitch)
    int dtype;           // DT_F32 / DT_BF16 inputs (partials and accumulation are always fp32)
    int algo;            // as ConvArgs::algo
    int wgrad8;          // partials were sized for wgrad8_kernel's block shape (wgrad8_shape): use it
    int xcd;             // as ConvArgs::xcd
    int band;            // as ConvArgs::band
    const float* amax_g; const float* amax_x0; const float* amax_x1;     // algo 2 only: see ConvArgs
};

int launch_wgrad(const WgradArgs& a, int mode, hipStream_t st);
// block shape (COB out-channels x JB in-channels, TH-row tiles) wgrad8_kernel uses for a 3x3 layer under the three-piece scheme;
// false if the layer stays on wgrad_kernel
bool wgrad8_shape(int CA, int CBp, int& COB, int& JB, int& TH, int& TWo, bool bf16 = false);
// spatial tiles wgrad8_kernel walks for a layer (TH x TWo tiles over the virtual-row strip of the batch); 0 if the layer is not on wgrad8_kernel
int wgrad8_ntiles(int CA, int CBp, int N, int H, int W, bool bf16);
// transposed convs' weight gradient on the 8-wave kernel (wgradt8_kernel: fp32 three-piece scheme, 128 x 64 blocks, 8 x 8-pixel tiles per image)
bool wgradt8_takes(int CA, int CB);
int wgradt8_ntiles(int N, int H, int W);
// out[(i*CBr + j)*T + tap] = sum_s part[s][tap][i][j]   (OIHW / (Cin,Cout,2,2) layouts);  bgrad[i] = sum_s bpart[s][i]
int launch_wgrad_reduce(const float* part, const float* bpart, float* wgrad, float* bgrad, int psplit, int T, int CA,
                        int CBp, int CBr, hipStream_t st, int bias_n = 0);      // bias_n > 0: bpart is [psplit][bias_n] (default [psplit][CA])
