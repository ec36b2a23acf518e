/* eld_amd.h -- C ABI of libeld_amd.so: the MI355X (gfx950) hot path of ELD.
 *
 * The reference (Vandermode/ELD) has no FFI: its boundary for this path is three Python
 * duck-typed plugin points.  The Python package `eld_amd` implements those
 * plugin points and binds THIS header with ctypes; every entry point below names the reference
 * code it replaces (paths relative to the reference checkout).
 *
 * Conventions (all entry points):
 *   - plain C types only; device pointers are BORROWED (the caller, normally PyTorch's caching
 *     allocator, owns every allocation; scratch is passed in as `ws`);
 *   - asynchronous on `stream` (a hipStream_t passed as void*; NULL = default stream);
 *     no hipMalloc / hipFree / hipDeviceSynchronize inside, so calls are hipGraph-capturable;
 *   - return value is a hipError_t as int (0 = success) or a negative ELD_E* code; nothing throws;
 *   - re-entrant per stream; no global mutable state.
 */
#ifndef ELD_AMD_H
#define ELD_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ELD_ABI_VERSION 1

/* negative = argument errors (hipError_t values are >= 0) */
#define ELD_EINVAL   (-1)   /* bad shape / flag combination / null pointer                   */
#define ELD_ENOTSUP  (-2)   /* valid request this build does not implement                    */
#define ELD_EWS      (-3)   /* workspace too small                                             */

/* ---- noise-model terms: letters of NoiseModel(model=...) (noise.py:158-166, 175) ----------- */
#define ELD_SHOT_POISSON  1u   /* 'P'  z = Poisson(y/K)*K                      noise.py:158-159 */
#define ELD_SHOT_GAUSS    2u   /* 'p'  z = y + N*sqrt(max(K*y,1e-10))          noise.py:160-161 */
#define ELD_READ_GAUSS    4u   /* 'g'  z += N*max(g_scale,1e-10)               noise.py:165-166 */
#define ELD_READ_TL       8u   /* 'G'  Tukey-lambda read noise   [withheld from the reference:  */
#define ELD_ROW          16u   /* 'R'  per-sensor-row Gaussian    README.md:41, noise.py:173;   */
#define ELD_QUANT        32u   /* 'U'  uniform quantisation noise  follows the ELD paper and    */
#define ELD_CBIAS        64u   /* 'B'  per-channel colour bias     camera_params/release/ npy tables] */
#define ELD_CLIP        128u   /* fuse the caller's clip to [0,1]       dataset/sid_dataset.py:277 */

/* input element types */
#define ELD_IN_F32  0   /* float32 in [0,1]                                                      */
#define ELD_IN_U16  1   /* uint16 LMDB code, decoded as clip(u16/65535,0,1) (lmdb_dataset.py:38-39) */

/* Per-image parameter record: the tuple returned by NoiseModel._sample_params (noise.py:225)
 * plus the withheld-model terms.  64 bytes. */
typedef struct EldNoiseParams {
    float K;            /* system gain (ADU per e-)                    noise.py:220           */
    float g_scale;      /* Gaussian read-noise std (ADU)               noise.py:221           */
    float tl_lambda;    /* Tukey-lambda shape        ('G_shape')                              */
    float tl_scale;     /* Tukey-lambda scale (ADU)  ('G_scale' regression)                   */
    float row_scale;    /* row-noise std (ADU)       ('R_scale' regression)                   */
    float q_step;       /* quantisation step (ADU), 1                                         */
    float saturation;   /* 16383-800                                   noise.py:205           */
    float ratio;        /* exposure ratio                              noise.py:223           */
    float color_bias[4];/* per packed channel (ADU)  ('color_bias')                           */
    uint32_t sample_id_lo, sample_id_hi;  /* GLOBAL sample index -> Philox counter words 1,2  */
    uint32_t reserved[2];
} EldNoiseParams;

/* Variate planes of the debug/inject buffers: float[ELD_NPLANES][N*C*H*W]. */
#define ELD_PLANE_COUNT   0   /* Poisson count (as float)        */
#define ELD_PLANE_NSHOT   1   /* N(0,1) of the 'p' term          */
#define ELD_PLANE_NREAD   2   /* N(0,1) of the 'g' term          */
#define ELD_PLANE_TL      3   /* unit-scale Tukey-lambda variate */
#define ELD_PLANE_NROW    4   /* row normal, broadcast per pixel */
#define ELD_PLANE_UQ      5   /* quantisation uniform in [0,1)   */
#define ELD_NPLANES       6

int eld_abi_version(void);
const char* eld_build_info(void);               /* "gfx950 hipcc <ver> ..." */
const char* eld_error_string(int code);

/* Fused per-pixel noise sampler.  Replaces NoiseModelBase.__call__ (noise.py:149-170), batched:
 *   in     N*C*H*W elements, NCHW contiguous (packed raw: C=4 Bayer planes), type `in_dtype`
 *   out    float32, same shape.  NOT clipped unless ELD_CLIP (the reference's callers clip).
 *   params device array of N records
 *   seed   Philox key; counters come from (element index, params[n].sample_id), so the output
 *          does not depend on launch geometry or on how images are spread over GPUs
 *   inject optional float[ELD_NPLANES][numel]: take the variates from here instead of Philox
 *          (deterministic-arithmetic parity against the reference's own draws)
 *   dump   optional float[ELD_NPLANES][numel]: also write the variates that were used
 * ELD_ROW requires C == 4 (Bayer packing: channels 0,1 <- sensor row 2h, 2,3 <- 2h+1; noise.py:16-19). */
int eld_noise_forward(const void* in, int in_dtype, float* out, const EldNoiseParams* params,
                      int N, int C, int H, int W, uint32_t flags, uint64_t seed,
                      const float* inject, float* dump, void* stream);

/* Raw Philox4x32-10 words of the sampler's counter layout, for bit-exact RNG tests:
 * out[i*4..i*4+3] = philox(ctr=(index0+i, sample_id, stream|iter<<8), key=seed). */
int eld_philox_words(uint32_t* out, uint32_t n, uint32_t index0, uint64_t sample_id,
                     uint32_t stream, uint32_t iter, uint64_t seed, void* stream_h);

/* Bayer pack / unpack.  Replaces RawPacker.pack_raw_bayer / unpack_raw_bayer (noise.py:10-20,66-81),
 * batched: mosaic float32 [N,2h,2w] <-> packed float32 [N,4,h,w]. */
int eld_pack_bayer(const float* mosaic, float* packed, int N, int h, int w, void* stream);
int eld_unpack_bayer(const float* packed, float* mosaic, int N, int h, int w, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ELD_AMD_H */
