// unet_misc.h -- launchers of the HBM-bound U-Net step kernels (internal).
#pragma once
#include "conv.h"

enum PackKind { PACK_CONV_FWD = 0, PACK_CONV_BWD = 1, PACK_CONVT_FWD = 2, PACK_CONVT_BWD = 3 };

int launch_nchw_to_nhwc16(const float* x, float* y, int N, int C, int H, int W, hipStream_t st);
int launch_maxpool_fwd(const float* in, float* out, int N, int Ho, int Wo, int C, hipStream_t st);
int launch_maxpool_bwd(const float* act, const float* dp, const float* skip, float* g, int N, int Ho, int Wo, int C, hipStream_t st);
int launch_maxpool_bwd_codes(const unsigned* pool_codes, const unsigned* slope_codes, const float* dp, const float* skip, float* g, int N, int Ho, int Wo, int C,
                             hipStream_t st);
int launch_head_fwd(const float* in, const float* w, const float* b, float* out, int N, int H, int W, int OC, hipStream_t st);
size_t head_bwd_ws_floats();
int launch_head_bwd(const float* dout, const float* act, const float* w, float* g, float* dw, float* db, float* part,
                    int N, int H, int W, int OC, hipStream_t st);
// fused training head (forward + loss + backward in one pass over conv9_2's output): see unet_misc.hip
size_t head_train_ws_floats();
int launch_head_train(const void* act, int bf16, const float* w, const float* b, const float* tgt, float* out, void* g, float* part, float* loss,
                      int N, int H, int W, int OC, int mse, float grad_scale, hipStream_t st);
int launch_head_train_reduce(const float* part, float* dw, float* db, int N, int H, int W, int OC, int bf16, hipStream_t st);
size_t colsum_ws_floats(int C);
int launch_colsum(const float* x, float* out, float* part, size_t P, int C, hipStream_t st);
int launch_colsum_reduce(const float* part, float* out, int nblocks, int C, hipStream_t st);
int launch_pack(const float* src, float* dst, int kind, int Cout, int Cin, int Cinp, int T, hipStream_t st, int x3bn = 0);
struct PackJob { size_t src_off, dst_off; int kind, Cout, Cin, Cinp, T, first_block, bf16, amax_slot, x3bn, bfdbn, bfgbn; };   // bfgbn: 0 or the transposed-conv slab BN (conv_bfg.hip)   // bfdbn: 0 or the bf16 DMA slab BN (conv_bfd.hip)   // x3bn: 0 or the slab BN (pre-split slab layout)   // dst_off in floats; bf16: write bf16_t
struct PackJobs { int n; PackJob job[48]; };      // both directions of all 22 packed layers fit one launch (3 KB of kernel arguments)
int launch_pack_all(PackJobs& jobs, const float* params, float* ws, hipStream_t st, float* amax = nullptr);      // amax: per-layer max|w| slots
int launch_absmax(const float* x, size_t n, float* slot, hipStream_t st);
size_t l1_ws_floats();
int launch_l1(const float* out, const float* tgt, float* dout, float* loss, float* part, size_t n, float grad_scale, hipStream_t st);
int launch_mse(const float* out, const float* tgt, float* dout, float* loss, float* part, size_t n, float grad_scale, hipStream_t st);
int launch_adam(float* p, const float* g, float* m, float* v, size_t n, double lr, double b1, double b2, double eps, double wd,
                int step, double gscale, hipStream_t st);

// first layer (conv_first.hip): NCHW input with Cin <= 4 -> NHWC 32 channels
int launch_conv_first_fwd(const float* x, const float* w, const float* bias, float* out, int N, int Cin, int H, int W, int lrelu, hipStream_t st, unsigned* codes = nullptr);      // codes: slope codes of the output (conv.h ConvArgs::codes_out), honoured when conv_first_writes_codes(Cin)
bool conv_first_writes_codes(int Cin);
size_t conv_first_wgrad_ws_floats();
int launch_conv_first_wgrad(const float* g, const float* x, float* dw, float* db, float* part, int N, int Cin, int H, int W, hipStream_t st, bool x3 = false);

// bf16 activations (forward / inference path)
int launch_maxpool_fwd_bf16(const bf16_t* in, bf16_t* out, int N, int Ho, int Wo, int C, hipStream_t st);
int launch_head_fwd_bf16(const bf16_t* in, const float* w, const float* b, float* out, int N, int H, int W, int OC, hipStream_t st);
int launch_conv_first_fwd_bf16(const float* x, const float* w, const float* bias, bf16_t* out, int N, int Cin, int H, int W, int lrelu, hipStream_t st, unsigned* codes = nullptr);
int launch_maxpool_bwd_bf16(const bf16_t* act, const bf16_t* dp, const bf16_t* skip, bf16_t* g, int N, int Ho, int Wo, int C, hipStream_t st);
int launch_head_bwd_bf16(const float* dout, const bf16_t* act, const float* w, bf16_t* g, float* dw, float* db, float* part,
                         int N, int H, int W, int OC, hipStream_t st);
int launch_colsum_bf16(const bf16_t* x, float* out, float* part, size_t P, int C, hipStream_t st);
int launch_conv_first_wgrad_bf16(const bf16_t* g, const float* x, float* dw, float* db, float* part, int N, int Cin, int H, int W, hipStream_t st);
