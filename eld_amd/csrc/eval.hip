// eval.hip -- evaluation-side kernels of the path, all HBM-bound single passes:
//   * quality_assess: PSNR and multichannel SSIM of util/index.py:76-81 on the x255-clipped images of tensor2im
//     (models/ELD_model.py:23-38), i.e. skimage.metrics.peak_signal_noise_ratio / structural_similarity(data_range=255,
//     multichannel=True) with skimage's defaults (7x7 uniform window, K1 = 0.01, K2 = 0.03, sample covariance, windows
//     fully inside the image, channels averaged) -- the frames never leave the device;
//   * illuminance_correct: models/ELD_model.py:138-169, out = <p,s>/<p,p> * p over the elements with s != 1, p clamped
//     to [0,1], one scale per image.
// Sums are accumulated in double and reduced in a fixed order (run-to-run bit-stable, no atomics).
#include "common.h"

namespace {

constexpr int QA_TW = 32, QA_TH = 8, WIN = 7, HALO = WIN - 1;
constexpr int RED_BLOCKS = 512;

// tensor2im (ELD_model.py:23-38): np.clip(x * 255, 0, 255).  mul = 255 for [0,1] inputs; mul = 1 (an exact no-op multiply) for values
// that are already images on the [0, range] scale (util/index.py's own callers)
__device__ __forceinline__ float to_im(float v, float mul, float range) { return fminf(fmaxf(v * mul, 0.f), range); }

__device__ __forceinline__ double block_sum(double v, double* sh) {       // 256 threads, fixed tree
    const int t = threadIdx.x;
    sh[t] = v;
    __syncthreads();
#pragma unroll
    for (int s = 128; s > 0; s >>= 1) {
        if (t < s) sh[t] += sh[t + s];
        __syncthreads();
    }
    const double r = sh[0];
    __syncthreads();
    return r;
}

// one workgroup = one 32x8 tile of window positions of one (image, channel) plane; writes the tile's sum of S
__global__ __launch_bounds__(256) void ssim_kernel(const float* __restrict__ est, const float* __restrict__ ref, double* __restrict__ part,
                                                   int H, int W, int tiles_x, int tiles_y, float mul, float scale) {
    __shared__ float lx[QA_TH + HALO][QA_TW + HALO], ly[QA_TH + HALO][QA_TW + HALO];
    __shared__ double hs[5][QA_TH + HALO][QA_TW];
    __shared__ double red[256];
    const int plane = blockIdx.y;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int y0 = ty * QA_TH, x0 = tx * QA_TW;
    const int Ho = H - HALO, Wo = W - HALO;                       // window positions
    const float* px = est + (size_t)plane * H * W;
    const float* py = ref + (size_t)plane * H * W;
    for (int i = threadIdx.x; i < (QA_TH + HALO) * (QA_TW + HALO); i += 256) {
        const int r = i / (QA_TW + HALO), c = i - r * (QA_TW + HALO);
        const int gy = y0 + r, gx = x0 + c;
        const bool ok = gy < H && gx < W;
        lx[r][c] = ok ? to_im(px[(size_t)gy * W + gx], mul, scale) : 0.f;
        ly[r][c] = ok ? to_im(py[(size_t)gy * W + gx], mul, scale) : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < (QA_TH + HALO) * QA_TW; i += 256) {      // horizontal 7-sums of x, y, xx, yy, xy
        const int r = i / QA_TW, c = i - r * QA_TW;
        double sx = 0, sy = 0, sxx = 0, syy = 0, sxy = 0;
#pragma unroll
        for (int d = 0; d < WIN; ++d) {
            const double a = lx[r][c + d], b = ly[r][c + d];
            sx += a; sy += b; sxx += a * a; syy += b * b; sxy += a * b;
        }
        hs[0][r][c] = sx; hs[1][r][c] = sy; hs[2][r][c] = sxx; hs[3][r][c] = syy; hs[4][r][c] = sxy;
    }
    __syncthreads();
    const int r = threadIdx.x / QA_TW, c = threadIdx.x - r * QA_TW;
    double S = 0.0;
    if (y0 + r < Ho && x0 + c < Wo) {
        double s[5];
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            double a = 0;
#pragma unroll
            for (int d = 0; d < WIN; ++d) a += hs[q][r + d][c];
            s[q] = a * (1.0 / (WIN * WIN));
        }
        const double cov_norm = (double)(WIN * WIN) / (WIN * WIN - 1.0);
        const double ux = s[0], uy = s[1];
        const double vx = cov_norm * (s[2] - ux * ux), vy = cov_norm * (s[3] - uy * uy), vxy = cov_norm * (s[4] - ux * uy);
        const double C1 = (0.01 * (double)scale) * (0.01 * (double)scale), C2 = (0.03 * (double)scale) * (0.03 * (double)scale);
        S = ((2 * ux * uy + C1) * (2 * vxy + C2)) / ((ux * ux + uy * uy + C1) * (vx + vy + C2));
    }
    const double tot = block_sum(S, red);
    if (threadIdx.x == 0) part[(size_t)plane * tiles_x * tiles_y + blockIdx.x] = tot;
}

// squared error of the x255-clipped images: grid (RED_BLOCKS, N)
__global__ __launch_bounds__(256) void sqerr_kernel(const float* __restrict__ est, const float* __restrict__ ref, double* __restrict__ part, size_t chw, float mul, float scale) {
    __shared__ double red[256];
    const float* a = est + (size_t)blockIdx.y * chw;
    const float* b = ref + (size_t)blockIdx.y * chw;
    double s = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < chw; i += (size_t)RED_BLOCKS * 256) {
        const double d = (double)to_im(b[i], mul, scale) - (double)to_im(a[i], mul, scale);
        s += d * d;
    }
    const double tot = block_sum(s, red);
    if (threadIdx.x == 0) part[(size_t)blockIdx.y * RED_BLOCKS + blockIdx.x] = tot;
}

// one workgroup per image: out[2n] = PSNR, out[2n+1] = SSIM
__global__ __launch_bounds__(256) void qa_final_kernel(const double* __restrict__ sq, const double* __restrict__ ss, double* __restrict__ out, int C,
                                                       int tiles, size_t chw, size_t windows, float scale) {
    __shared__ double red[256];
    const int n = blockIdx.x;
    double a = 0.0;
    for (int i = threadIdx.x; i < RED_BLOCKS; i += 256) a += sq[(size_t)n * RED_BLOCKS + i];
    const double err = block_sum(a, red) / (double)chw;
    double ssim = 0.0;
    for (int c = 0; c < C; ++c) {                      // mean over channels of the per-channel means (multichannel=True)
        double b = 0.0;
        for (int i = threadIdx.x; i < tiles; i += 256) b += ss[((size_t)n * C + c) * tiles + i];
        ssim += block_sum(b, red) / (double)windows;
    }
    if (threadIdx.x == 0) {
        out[2 * n] = 10.0 * log10((double)scale * (double)scale / err);
        out[2 * n + 1] = ssim / C;
    }
}

// illuminance correction, pass 1: per-block partial <p,s> and <p,p> over s != 1 (p clamped to [0,1]); grid (RED_BLOCKS, N)
__global__ __launch_bounds__(256) void illum_dot_kernel(const float* __restrict__ pred, const float* __restrict__ src, double* __restrict__ part, size_t chw,
                                                        size_t src_stride) {
    __shared__ double red[256];
    const float* p = pred + (size_t)blockIdx.y * chw;
    const float* s = src + (size_t)blockIdx.y * src_stride;
    double num = 0.0, den = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < chw; i += (size_t)RED_BLOCKS * 256) {
        const float sv = s[i];
        if (sv != 1.0f) {
            const double pv = fminf(fmaxf(p[i], 0.f), 1.f);
            num += pv * (double)sv; den += pv * pv;
        }
    }
    const double tn = block_sum(num, red), td = block_sum(den, red);
    if (threadIdx.x == 0) { part[((size_t)blockIdx.y * RED_BLOCKS + blockIdx.x) * 2] = tn; part[((size_t)blockIdx.y * RED_BLOCKS + blockIdx.x) * 2 + 1] = td; }
}

__global__ __launch_bounds__(256) void illum_alpha_kernel(const double* __restrict__ part, float* __restrict__ alpha) {
    __shared__ double red[256];
    const int n = blockIdx.x;
    double a = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < RED_BLOCKS; i += 256) { a += part[((size_t)n * RED_BLOCKS + i) * 2]; b += part[((size_t)n * RED_BLOCKS + i) * 2 + 1]; }
    const double num = block_sum(a, red), den = block_sum(b, red);
    if (threadIdx.x == 0) alpha[n] = (float)num / (float)den;             // num / den as fp32 tensors (ELD_model.py:164-166)
}

__global__ __launch_bounds__(256) void illum_scale_kernel(const float* __restrict__ pred, const float* __restrict__ alpha, float* __restrict__ out, size_t chw) {
    const float a = alpha[blockIdx.y];
    const size_t base = (size_t)blockIdx.y * chw;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < chw; i += (size_t)gridDim.x * 256)
        out[base + i] = a * fminf(fmaxf(pred[base + i], 0.f), 1.f);
}

// ---- raw -> sRGB ISP: util/process.py:52-68 `process`, one pass: 16 B read + 12 B written per RGBG position.
//      gains (15-19) -> clamp -> binning R, (G1+G2)/2, B (42-49) -> 3x3 CCM with j ascending (22-31) -> clamp -> gamma
//      compression max(x,1e-8)^(1/gamma) (34-39) or piecewise-linear camera response (71-83) -> truncating 8-bit quantiser.
//      gamma == 2.2 (the only value the reference passes): power + quantiser are ONE exact step function of the clamped input,
//      tabulated from torch's own float32 evaluation over every float32 in [0, 1] (gamma22_table.h, oracle/gen_gamma_table.py):
//      bit-exact 8-bit codes.  Other gammas: the power in double, rounded once (then +-1 code on isolated pixels is possible).
#include "gamma22_table.h"
__device__ __forceinline__ float isp_quant(float v) {
    int q = (int)(v * 255.0f);                                     // .int(): truncation toward zero
    q = q < 0 ? 0 : (q > 255 ? 255 : q);
    return (float)q / 255.0f;
}

__global__ __launch_bounds__(256) void isp_kernel(const float* __restrict__ bayer, const float* __restrict__ wbs, const float* __restrict__ ccms,
                                                  float* __restrict__ out, size_t hw, float inv_gamma, const float* __restrict__ crf_E,
                                                  const float* __restrict__ crf_f, int crf_n, int gtab) {
    __shared__ unsigned s_t[256];
    if (gtab) s_t[threadIdx.x] = ELD_GAMMA22_T[threadIdx.x];
    __syncthreads();
    const int n = blockIdx.y;
    const float* wb = wbs + 4 * n;
    const float* cm = ccms + 9 * n;
    const float w0 = wb[0], w1 = wb[1], w2 = wb[2], w3 = wb[3];
    float m[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) m[i] = cm[i];
    const float* src = bayer + (size_t)n * 4 * hw;
    float* dst = out + (size_t)n * 3 * hw;
    const double ig = (double)inv_gamma;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < hw; i += (size_t)gridDim.x * 256) {
        const float c0 = fminf(fmaxf(src[i] * w0, 0.f), 1.f), c1 = fminf(fmaxf(src[hw + i] * w1, 0.f), 1.f);
        const float c2 = fminf(fmaxf(src[2 * hw + i] * w2, 0.f), 1.f), c3 = fminf(fmaxf(src[3 * hw + i] * w3, 0.f), 1.f);
        const float r = c0, g = (c1 + c3) / 2.0f, b = c2;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float v = r * m[3 * c];
            v = v + g * m[3 * c + 1];
            v = v + b * m[3 * c + 2];
            v = fminf(fmaxf(v, 0.f), 1.f);
            float o;
            if (crf_n > 0) {
                int lo = 0, hi = crf_n;                              // searchsorted(E, v, 'left') - 1, clamped to [0, n-2]
                while (lo < hi) { const int mid = (lo + hi) >> 1; if (crf_E[mid] < v) lo = mid + 1; else hi = mid; }
                int ind = lo - 1;
                ind = ind < 0 ? 0 : (ind > crf_n - 2 ? crf_n - 2 : ind);
                const float slope = (crf_f[ind + 1] - crf_f[ind]) / (crf_E[ind + 1] - crf_E[ind]);
                o = crf_f[ind] + slope * (v - crf_E[ind]);
            } else if (gtab) {                                      // exact: code = #{c : bits(max(v,1e-8)) >= T[c]}, found from a hardware-pow guess
                const float vm = fmaxf(v, 1e-8f);
                const unsigned vb = __float_as_uint(vm);
                int q = (int)(__builtin_amdgcn_exp2f(__builtin_amdgcn_logf(vm) * inv_gamma) * 255.0f);
                q = q < 0 ? 0 : (q > 255 ? 255 : q);
                while (q < 255 && vb >= s_t[q + 1]) ++q;
                while (q > 0 && vb < s_t[q]) --q;
                dst[c * hw + i] = (float)q / 255.0f;
                continue;
            } else {
                o = (float)pow((double)fmaxf(v, 1e-8f), ig);
            }
            dst[c * hw + i] = isp_quant(o);
        }
    }
}

inline size_t qa_tiles(int H, int W) { return (size_t)((W - HALO + QA_TW - 1) / QA_TW) * ((H - HALO + QA_TH - 1) / QA_TH); }

}  // namespace

extern "C" size_t eld_quality_assess_workspace_bytes(int N, int C, int H, int W) {
    if (N < 1 || C < 1 || H < WIN || W < WIN) return 0;
    return ((size_t)N * RED_BLOCKS + (size_t)N * C * qa_tiles(H, W)) * sizeof(double);
}

static int quality_assess_impl(const float* est, const float* ref, double* out, void* ws, size_t ws_bytes, int N, int C, int H, int W,
                               float mul, float data_range, void* stream) {
    if (N == 0) return 0;
    if (!est || !ref || !out || !ws || N < 0 || C < 1 || H < WIN || W < WIN || !(data_range > 0.f)) return ELD_EINVAL;
    if (ws_bytes < eld_quality_assess_workspace_bytes(N, C, H, W)) return ELD_EWS;
    hipStream_t st = as_stream(stream);
    double* sq = (double*)ws;
    double* ss = sq + (size_t)N * RED_BLOCKS;
    const int tiles_x = (W - HALO + QA_TW - 1) / QA_TW, tiles_y = (H - HALO + QA_TH - 1) / QA_TH;
    const size_t chw = (size_t)C * H * W;
    ELD_LAUNCH(sqerr_kernel, dim3(RED_BLOCKS, N), dim3(256), 0, st, est, ref, sq, chw, mul, data_range);
    ELD_LAUNCH_CHECK();
    ELD_LAUNCH(ssim_kernel, dim3(tiles_x * tiles_y, N * C), dim3(256), 0, st, est, ref, ss, H, W, tiles_x, tiles_y, mul, data_range);
    ELD_LAUNCH_CHECK();
    ELD_LAUNCH(qa_final_kernel, dim3(N), dim3(256), 0, st, sq, ss, out, C, tiles_x * tiles_y, chw, (size_t)(H - HALO) * (W - HALO), data_range);
    ELD_LAUNCH_CHECK();
    return 0;
}

extern "C" int eld_quality_assess(const float* est, const float* ref, double* out, void* ws, size_t ws_bytes, int N, int C, int H, int W,
                                  float data_range, void* stream) {
    return quality_assess_impl(est, ref, out, ws, ws_bytes, N, C, H, W, data_range, data_range, stream);
}

// the same on images that are ALREADY on the [0, data_range] scale (util/index.py:76-81 as its callers use it: quality_assess(X, Y) on
// tensor2im outputs): no x255 stage, values only clipped to [0, data_range] (a no-op for tensor2im outputs)
extern "C" int eld_quality_assess_images(const float* est, const float* ref, double* out, void* ws, size_t ws_bytes, int N, int C, int H, int W,
                                         float data_range, void* stream) {
    return quality_assess_impl(est, ref, out, ws, ws_bytes, N, C, H, W, 1.0f, data_range, stream);
}

extern "C" size_t eld_illuminance_correct_workspace_bytes(int N) { return N < 1 ? 0 : (size_t)N * RED_BLOCKS * 2 * sizeof(double) + (size_t)N * sizeof(float); }

extern "C" int eld_illuminance_correct(const float* predict, const float* source, float* out, void* ws, size_t ws_bytes, int N, int source_N,
                                       size_t chw, void* stream) {
    if (N == 0) return 0;
    if (!predict || !source || !out || !ws || N < 0 || chw == 0 || (source_N != N && source_N != 1)) return ELD_EINVAL;
    if (ws_bytes < eld_illuminance_correct_workspace_bytes(N)) return ELD_EWS;
    hipStream_t st = as_stream(stream);
    double* part = (double*)ws;
    float* alpha = (float*)(part + (size_t)N * RED_BLOCKS * 2);
    ELD_LAUNCH(illum_dot_kernel, dim3(RED_BLOCKS, N), dim3(256), 0, st, predict, source, part, chw, source_N == 1 ? (size_t)0 : chw);
    ELD_LAUNCH_CHECK();
    ELD_LAUNCH(illum_alpha_kernel, dim3(N), dim3(256), 0, st, part, alpha);
    ELD_LAUNCH_CHECK();
    ELD_LAUNCH(illum_scale_kernel, dim3(1024, N), dim3(256), 0, st, predict, alpha, out, chw);
    ELD_LAUNCH_CHECK();
    return 0;
}

extern "C" int eld_isp_process(const float* bayer, const float* wbs, const float* ccms, float* out, int N, int H, int W, float gamma,
                               const float* crf_E, const float* crf_f, int crf_n, void* stream) {
    if (N == 0) return 0;
    if (!bayer || !wbs || !ccms || !out || N < 0 || H < 1 || W < 1 || !(gamma > 0.f)) return ELD_EINVAL;
    if (crf_n != 0 && (crf_n < 2 || !crf_E || !crf_f)) return ELD_EINVAL;
    const size_t hw = (size_t)H * W;
    const unsigned bx = (unsigned)((hw + 255) / 256 < 2048 ? (hw + 255) / 256 : 2048);
    const int gtab = (crf_n == 0 && gamma == 2.2f) ? 1 : 0;
    ELD_LAUNCH(isp_kernel, dim3(bx, N), dim3(256), 0, as_stream(stream), bayer, wbs, ccms, out, hw, (float)(1.0 / (double)gamma), crf_E, crf_f, crf_n, gtab);
    ELD_LAUNCH_CHECK();
    return 0;
}
