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
 *   - re-entrant per stream.  Process-wide state is limited to (a) immutable per-device caches (compute-unit count,
 *     per-kernel LDS attribute set once per device) and (b) the DEFAULT fp32 product scheme of eld_conv_fp32_algo(), which
 *     only the entry points that do not take a scheme argument consult; eld_unet_forward_ex / eld_unet_backward_ex name the
 *     scheme per call and never read it.  Developer switches (ELD_CONV_DBG, ELD_NOISE_DBG, eld_debug_conv_prof) exist only in
 *     builds made with -DELD_DEV_TOOLS=1; the default build ignores them.
 */
#ifndef ELD_AMD_H
#define ELD_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2 (round 6): eld_unet_infer_ex and eld_debug_ws_state_entries exist; the U-Net workspace grew (slope-code regions) -- size it with
 * eld_unet_workspace_bytes of the SAME library; eld_unet_backward_ex accepts an explicit dout after eld_unet_forward_loss_ex and refuses a backward
 * after eld_unet_infer_ex.  A binding must compare eld_abi_version() with the ELD_ABI_VERSION it was written against. */
#define ELD_ABI_VERSION 2

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
#define ELD_AUG_NOTRANSPOSE 256u /* eld_augment only: no image of the batch has its transpose bit set (then H != W is fine) */

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

/* The same sampler with explicit image strides (in ELEMENTS): image n is read at in + n*in_image_stride and written at
 * out + n*out_image_stride.  Burst synthesis (SynDataset, dataset/sid_dataset.py:267-273: num_burst noisy frames of ONE clean
 * image with ONE parameter draw, concatenated on the channel axis) is num_burst launches with out_image_stride =
 * num_burst*C*H*W and out advanced by k*C*H*W, each with its own sample ids; in_image_stride 0 re-reads one clean image. */
int eld_noise_forward_strided(const void* in, int in_dtype, size_t in_image_stride, float* out, size_t out_image_stride,
                              const EldNoiseParams* params, int N, int C, int H, int W, uint32_t flags, uint64_t seed,
                              const float* inject, float* dump, void* stream);

/* Rounds of the Philox4x32 generator this build's sampler runs (7 since round 3 of this library; -DELD_PHILOX_ROUNDS=10 restores
 * cuRAND's count).  The noise stream of a (seed, sample id) pair is a function of this number: a binding that pins or replays streams
 * (checkpoints, golden vectors, the test oracle) must compare it with the count it was made for -- eld_amd/_lib.py does at load time. */
int eld_philox_rounds(void);

/* Raw Philox4x32-7 words (ELD_PHILOX_ROUNDS, csrc/philox.h) of the sampler's counter layout, for bit-exact RNG tests:
 * out[i*4..i*4+3] = philox(ctr=(index0+i, sample_id, stream|iter<<8), key=seed). */
int eld_philox_words(uint32_t* out, uint32_t n, uint32_t index0, uint64_t sample_id,
                     uint32_t stream, uint32_t iter, uint64_t seed, void* stream_h);

/* Bayer pack / unpack.  Replaces RawPacker.pack_raw_bayer / unpack_raw_bayer (noise.py:10-20,66-81),
 * batched: mosaic float32 [N,2h,2w] <-> packed float32 [N,4,h,w]. */
int eld_pack_bayer(const float* mosaic, float* packed, int N, int h, int w, void* stream);
int eld_unpack_bayer(const float* packed, float* mosaic, int N, int h, int w, void* stream);
/* X-Trans pack / unpack.  Replaces RawPacker.pack_raw_xtrans / unpack_raw_xtrans (noise.py:22-64, 83-127), batched:
 * mosaic float32 [N,Hm,Wm] -> packed float32 [N,9,2*(Hm/6),2*(Wm/6)] (the reference truncates to whole 6x6 cells, noise.py:25-26);
 * packed float32 [N,9,h,w] -> mosaic float32 [N,3h,3w].  Index maps only: bit-exact. */
int eld_pack_xtrans(const float* mosaic, float* packed, int N, int Hm, int Wm, void* stream);
int eld_unpack_xtrans(const float* packed, float* mosaic, int N, int h, int w, void* stream);
/* pack_raw_bayer (dataset/sid_dataset.py:172-196): uint16 sensor mosaic [N,2h,2w] (raw.raw_image_visible) -> packed float32
 * [N,4,h,w] in the order R, G1, B, G2 given by the 2x2 `raw_pattern` (row-major colour codes 0..3, HOST array of 4 ints),
 * normalised per channel: clip((x - black_level[k]) / (white_point - black_level[k]), 0, 1), float32 arithmetic as NumPy's
 * (black_level: HOST array of 4 floats = raw.black_level_per_channel; white_point 16383 in the reference).  Bit-exact. */
int eld_pack_raw_bayer_u16(const uint16_t* mosaic, float* packed, int N, int h, int w, const int* raw_pattern,
                           const float* black_level, float white_point, void* stream);


/* Training-pair augmentation of ELDTrainDataset.__getitem__ (dataset/sid_dataset.py:344-352), batched on device:
 * per image n, bits of aug[n]: 1 = flip H (axis 1), 2 = flip W (axis 2), 4 = transpose (0,2,1), applied in that order;
 * ELD_CLIP in `flags` fuses the clip to [0,1] of sid_dataset.py:354.  A transposed image needs H == W (batched tensor): with
 * H != W the call returns ELD_ENOTSUP unless ELD_AUG_NOTRANSPOSE is set, which makes the kernel ignore bit 4.
 * in/out: float32 [N,C,H,W]; aug: device int32[N].  Pure index map: bit-exact. */
int eld_augment(const float* in, float* out, const int32_t* aug, int N, int C, int H, int W, uint32_t flags, void* stream);
/* The same on uint16 LMDB codes: out = augment(clip(u16/65535, 0, 1)) -- LMDBDataset.__getitem__'s decode (dataset/lmdb_dataset.py:
 * 35-39, true fp32 division: bit-exact for all 65536 codes) fused in front; aug == NULL is the plain decode. */
int eld_augment_u16(const uint16_t* in, float* out, const int32_t* aug, int N, int C, int H, int W, uint32_t flags, void* stream);

/* ====================================================================================================
 * U-Net ("See-in-the-Dark", 5 scales) -- replaces UNetSeeInDark.forward (models/arch/Unet.py:48-91) and
 * the autograd backward that ELDModel.backward_G triggers (models/ELD_model.py:411-420).
 *
 * Parameters live in ONE flat float32 buffer, tensors in the reference's named_parameters() order
 * (conv1_1.weight, conv1_1.bias, ... conv5_2, upv6, conv6_1, conv6_2, ... upv9, conv9_1, conv9_2, conv10_1),
 * each in the reference's own layout (Conv2d OIHW, ConvTranspose2d (Cin,Cout,2,2)), so a state_dict
 * maps onto it by plain views (models/ELD_model.py:516-523) and a data-parallel gradient all-reduce is
 * one contiguous buffer.  Activations are kept NHWC float32 in the caller-provided workspace.
 * x / out / dout are NCHW float32 like the reference's tensors.  H and W must be multiples of 16.
 * ==================================================================================================== */
#define ELD_UNET_NTENSORS 46

/* offsets[i] = first float of tensor i in the flat buffer, offsets[46] = total count (7,760,484 for 4->4). */
int eld_unet_param_offsets(int in_ch, int out_ch, int64_t* offsets /* [ELD_UNET_NTENSORS+1] */);
/* bytes of scratch eld_unet_forward/backward need for this shape (packed weights, activations, gradients, partials) */
size_t eld_unet_workspace_bytes(int N, int H, int W, int in_ch, int out_ch);
int eld_unet_forward(const float* x, const float* params, float* out, void* ws, size_t ws_bytes,
                     int N, int H, int W, int in_ch, int out_ch, void* stream);
/* Inference in bf16 (BASELINE config 3's precision): bf16 NHWC activations and packed weights on v_mfma_f32_32x32x16_bf16,
 * fp32 accumulation, bias, first-layer input and output.  Same arguments and workspace as eld_unet_forward; the saved
 * activations are bf16: follow it with eld_unet_backward_bf16, never with eld_unet_backward. */
int eld_unet_forward_bf16(const float* x, const float* params, float* out, void* ws, size_t ws_bytes,
                          int N, int H, int W, int in_ch, int out_ch, void* stream);
/* Backward of a bf16 forward: bf16 activation gradients on the bf16 MFMA; parameter gradients are accumulated and
 * written in fp32 (fp32 master weights and Adam are unchanged).  Same contract as eld_unet_backward. */
int eld_unet_backward_bf16(const float* dout, const float* params, float* grads, void* ws, size_t ws_bytes,
                           int N, int H, int W, int in_ch, int out_ch, void* stream);
/* Needs the workspace exactly as eld_unet_forward left it (saved activations).  Writes every element of grads. */
int eld_unet_backward(const float* dout, const float* params, float* grads, void* ws, size_t ws_bytes,
                      int N, int H, int W, int in_ch, int out_ch, void* stream);

/* eld_unet_backward / eld_unet_backward_bf16 (precision 0 / 1) for data-parallel training (the reference is
 * single-device, models/ELD_model.py:187-190): the flat gradient buffer is cut into n_buckets contiguous buckets starting at the
 * ascending float offsets bucket_start[k] (bucket k = [bucket_start[k], bucket_start[k+1]) ; the last one runs to the end;
 * bucket_start[0] is normally 0).  Gradients are produced from the END of the buffer towards its start, and
 * hipEventRecord(bucket_event[k], stream) is enqueued as soon as the last kernel writing bucket k is enqueued, so a
 * communication stream can wait on the events and all-reduce bucket by bucket while the rest of the backward runs.
 * bucket_event[k]: hipEvent_t created by the caller. */
int eld_unet_backward_buckets(const float* dout, const float* params, float* grads, void* ws, size_t ws_bytes,
                              int N, int H, int W, int in_ch, int out_ch, int precision,
                              const int64_t* bucket_start, void* const* bucket_event, int n_buckets, void* stream);

/* Training forward with the loss fused into the head (ELD_model.py:469-475: forward() + backward_G()'s loss): as eld_unet_forward_ex with the
 * activations kept for the backward, but the last layer (conv10_1, Unet.py:46,88), the loss against `target` (loss_kind 0: nn.L1Loss, 1:
 * nn.MSELoss, models/losses.py:30-34; mean over all elements, written to *loss on the device) and the head's own backward run as ONE pass over
 * conv9_2's output: `out` is written, the output gradient never touches memory, the gradient of conv9_2's output and the head's weight-gradient
 * partials stay in the workspace.  Follow with eld_unet_backward_ex(dout = NULL, same workspace, same shape / precision), which finishes the
 * head's dW / db and runs the rest of the backward.  grad_scale multiplies dLoss/dout (1 for a plain mean loss).
 * The pair shares more than the head: the forward packs the weights for both directions in one launch (the backward differentiates at the
 * parameters the forward ran with), and the backward's first-layer weight gradient reads `x` where the forward read it -- no copy of the input is
 * kept in the workspace.  `x` (and `params`) must therefore stay valid and unchanged until the backward that follows this forward on the
 * workspace has been enqueued on the same stream -- the matching eld_unet_backward_ex(dout = NULL), or a backward with an explicit dout
 * (allowed: the head is then recomputed from dout, the weights are packed again, and the first layer's weight gradient still reads `x`
 * from the caller). */
int eld_unet_forward_loss_ex(const float* x, const float* params, const float* target, float* out, float* loss, void* ws, size_t ws_bytes,
                             int N, int H, int W, int in_ch, int out_ch, int precision, int fp32_algo, int loss_kind, float grad_scale, void* stream);
/* The same two calls with everything per call: precision 0 = fp32 / 1 = bf16 activations; fp32_algo names the fp32 product
 * scheme (see eld_conv_fp32_algo below; < 0 = the process default); n_buckets may be 0.  A backward must name the scheme its
 * forward ran with: scheme 2 leaves operand bounds in the workspace that only a scheme-2 backward reads.
 * eld_unet_backward_ex accepts dout == NULL when (and only when) the LAST forward on this workspace (host call order) was
 * eld_unet_forward_loss_ex with the same N / H / W / channels / precision; otherwise it returns ELD_EINVAL (the library remembers, per
 * workspace pointer, which forward filled it -- host bookkeeping, no device read). */
int eld_unet_forward_ex(const float* x, const float* params, float* out, void* ws, size_t ws_bytes,
                        int N, int H, int W, int in_ch, int out_ch, int precision, int fp32_algo, void* stream);
int eld_unet_backward_ex(const float* dout, const float* params, float* grads, void* ws, size_t ws_bytes,
                         int N, int H, int W, int in_ch, int out_ch, int precision, int fp32_algo,
                         const int64_t* bucket_start, void* const* bucket_event, int n_buckets, void* stream);
/* The forward under torch.no_grad() (ELD_model.py:203-307 eval / test: `self.netG(self.input)` with nothing kept): eld_unet_forward_ex's arguments and
 * bit-identical output, but nothing a backward would need is produced -- no copy of the input in the workspace, no slope codes (round 5: the training
 * forwards also write 2 bits per element of four activations for the backward-data epilogues).  eld_unet_backward_ex on a workspace whose last forward was
 * this call returns ELD_EINVAL. */
int eld_unet_infer_ex(const float* x, const float* params, float* out, void* ws, size_t ws_bytes,
                      int N, int H, int W, int in_ch, int out_ch, int precision, int fp32_algo, void* stream);

/* How the fp32 3x3 convolutions of eld_unet_forward/backward and eld_conv3x3_* form their products:
 *   0  v_mfma_f32_32x32x2_f32 (fp32 operands);
 *   1  every fp32 operand cut exactly into three bf16 pieces, six v_mfma_f32_32x32x16_bf16 per k-block, fp32 accumulate
 *      (same fp32-level accuracy, see csrc/conv_x3.hip);
 *   2  every fp32 operand scaled by a per-tensor power of two and cut into two fp16 pieces (22 significant bits), three
 *      v_mfma_f32_32x32x16_f16 per k-block, fp32 accumulate: each product is good to 2^-22 instead of 2^-24, which stays
 *      inside the fp32 dot-product error bound for every contraction length >= 4 (csrc/conv_igemm.hip, H2).
 * algo < 0 only queries.  Process-wide DEFAULT for the entry points without a scheme argument; returns the value in force
 * before the call.  Initial value: env ELD_FP32_CONV. */
int eld_conv_fp32_algo(int algo);

/* mean |out-target| (nn.L1Loss, models/losses.py:32) and, if dout != NULL, its gradient times grad_scale.
 * ws: eld_l1_workspace_bytes() bytes.  loss: one device float. */
size_t eld_l1_workspace_bytes(void);
int eld_l1_loss(const float* out, const float* target, float* dout, float* loss, void* ws, size_t n, float grad_scale, void* stream);
/* --loss l2: nn.MSELoss (models/losses.py:34), same contract and workspace as eld_l1_loss. */
int eld_mse_loss(const float* out, const float* target, float* dout, float* loss, void* ws, size_t n, float grad_scale, void* stream);
/* torch.optim.Adam step over a flat buffer (models/ELD_model.py:400-401,475); step counts from 1;
 * the gradient is multiplied by grad_scale first (1/world_size after a sum all-reduce). */
int eld_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, size_t n, double lr, double beta1,
                  double beta2, double eps, double weight_decay, int step, double grad_scale, void* stream);

/* ---- evaluation side -------------------------------------------------------------------------
 * util/index.py:76-81 quality_assess on the images of tensor2im (models/ELD_model.py:23-38: clip(x*255, 0, 255)):
 * est/ref are NCHW float32 in [0,1] units; out[2n] = PSNR (dB), out[2n+1] = SSIM of image n, as device doubles.
 * SSIM = skimage.metrics.structural_similarity(data_range=255, multichannel=True) with its defaults (7x7 uniform window,
 * K1=0.01, K2=0.03, sample covariance, windows inside the image, mean over channels); scikit-image is a third-party
 * dependency the reference does not pin. */
size_t eld_quality_assess_workspace_bytes(int N, int C, int H, int W);
int eld_quality_assess(const float* est, const float* ref, double* out, void* ws, size_t ws_bytes, int N, int C, int H, int W,
                       float data_range, void* stream);
/* the same for images already on the [0, data_range] scale (util/index.py:76-81 called on tensor2im outputs): no x255 stage */
int eld_quality_assess_images(const float* est, const float* ref, double* out, void* ws, size_t ws_bytes, int N, int C, int H, int W,
                              float data_range, void* stream);
/* models/ELD_model.py:138-169 IlluminanceCorrect: out[n] = <p,s>/<p,p> * p, p = clamp(predict[n], 0, 1), sums over the
 * elements with source != 1; source_N is N or 1 (one source for all).  chw = elements per image. */
size_t eld_illuminance_correct_workspace_bytes(int N);
int eld_illuminance_correct(const float* predict, const float* source, float* out, void* ws, size_t ws_bytes, int N, int source_N,
                            size_t chw, void* stream);

/* util/process.py:52-68 `process`: bayer (N,4,H,W) RGBG in [0,1] -> out (N,3,H,W) sRGB, quantised to
 * k/255.  wbs (N,4), ccms (N,3,3) row-major, all device float32.  crf_n = 0: gamma compression with 1/gamma; crf_n >= 2:
 * camera response by piecewise-linear interpolation of (crf_E, crf_f), ascending crf_E (torchinterp1d's rule). */
int eld_isp_process(const float* bayer, const float* wbs, const float* ccms, float* out, int N, int H, int W, float gamma,
                    const float* crf_E, const float* crf_f, int crf_n, void* stream);

/* Dev tool (tools/conv_phase_profile.py; a no-op unless built with -DELD_DEV_TOOLS=1): device buffer of 8 x 4 x 128 x 6 uint64 that conv_x3_kernel fills with s_memtime
 * stamps of its stage phases (first 8 workgroups, first 128 stages); NULL switches it off (default). */
void eld_debug_conv_prof(void* buf);
/* Test hook: route the bf16 launches that normally run on a specialised kernel back to the generic one, so that the parity tests can demand
 * BIT-IDENTICAL results from the two kernel families on the same inputs (same k order, same MFMA): mask bit 0 = conv_bfs_kernel (32-output-channel
 * 3x3 layers) off, bit 1 = conv_bfg_kernel (transposed convolutions) off, bit 2 = conv_bfd_kernel (the other bf16 3x3 layers) off, bit 3 = conv_bfw_kernel off (the 64-output-channel layers with K <= 64 then run on conv_bfd_kernel<64>; conv_bfw sums K in another order, so it equals the others up to fp32 summation order, not bit for bit), bit 4 = wgrad8d_kernel off (the bf16 weight gradient of the 128 x 64 blocks then runs the register-staged wgrad8_kernel<bf16>: other tile shape, so equal up to the fp32 summation order over pixels); bit 5 = eld_quality_assess with one window column per lane instead of two, bit 6 = eld_quality_assess on the round-2 tile kernels (three
 * implementations of the same sums: tests/test_model_gpu.py runs the oracle comparison under each); bit 7 = the U-Net forwards write no slope codes (the backward-data
 * epilogues of levels 0 / 1 then read the saved activations, as before round 5: same slopes, bit-identical gradients); bit 8 = no pool-argmax codes
 * (the backward of the two full-size pools then reads the saved un-pooled tensors: same winners, bit-identical gradients).  Returns the previous mask.  Process-wide; production never calls it. */
int eld_debug_kernel_mask(int mask);
/* Test hook: live entries of the per-workspace host bookkeeping (which forward last filled a workspace: fused head, slope codes); bounded, evicted
 * one least-recently-touched entry at a time. */
int eld_debug_ws_state_entries(void);

/* ---- single layers on NHWC float32 tensors with reference-layout weights; used by the parity tests ---- */
size_t eld_layer_workspace_bytes(int N, int H, int W, int Cin, int Cout);
/* out = [lrelu](conv3x3(cat[in0,in1]) + bias).  nn.Conv2d(k=3,p=1) + torch.max(0.2x,x)  (Unet.py:11-44,102-104) */
int eld_conv3x3_forward(const float* in0, int C0, const float* in1, int C1, const float* w_oihw, const float* bias,
                        float* out, int N, int H, int W, int Cout, int lrelu, void* ws, size_t ws_bytes, void* stream);
/* din = conv3x3_backward_data(g); channels [0,split) -> din0, [split,Cin) -> din1; optionally times the LeakyReLU
 * slope of the saved post-activation tensors act0/act1 (NULL = no activation in front). */
int eld_conv3x3_backward_data(const float* g, const float* w_oihw, float* din0, float* din1, int split, const float* act0,
                              const float* act1, int N, int H, int W, int Cin, int Cout, void* ws, size_t ws_bytes, void* stream);
/* dw (OIHW), db from g (grad of the pre-activation output) and the layer input cat[x0,x1]. */
int eld_conv3x3_backward_weight(const float* g, const float* x0, int C0, const float* x1, int C1, float* dw, float* db,
                                int N, int H, int W, int Cout, void* ws, size_t ws_bytes, void* stream);
/* nn.ConvTranspose2d(Cin,Cout,2,stride=2) (Unet.py:30,34,38,42): in [N,H,W,Cin] -> out [N,2H,2W,Cout]. */
int eld_convt2x2_forward(const float* in, const float* w, const float* bias, float* out, int N, int H, int W, int Cin, int Cout,
                         void* ws, size_t ws_bytes, void* stream);
int eld_convt2x2_backward_data(const float* dout, const float* w, const float* act, float* din, int N, int H, int W, int Cin,
                               int Cout, void* ws, size_t ws_bytes, void* stream);
int eld_convt2x2_backward_weight(const float* in, const float* dout, float* dw, float* db, int N, int H, int W, int Cin, int Cout,
                                 void* ws, size_t ws_bytes, void* stream);
/* nn.MaxPool2d(2) (Unet.py:13): in [N,2Ho,2Wo,C] -> out [N,Ho,Wo,C]; backward = (routed dp + skip) * slope(act). */
int eld_maxpool2x2_forward(const float* in, float* out, int N, int Ho, int Wo, int C, void* stream);
int eld_maxpool2x2_backward(const float* act, const float* dp, const float* skip, float* g, int N, int Ho, int Wo, int C, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ELD_AMD_H */
