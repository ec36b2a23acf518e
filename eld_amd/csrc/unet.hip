// unet.hip -- host-side orchestration of the SID U-Net forward / backward over the HIP kernels, and
// the C-ABI entry points of include/eld_amd.h for it.  No allocation, no synchronisation: every launch
// goes to the caller's stream and all scratch comes out of the caller's workspace, so a whole training
// step is one stream-ordered chain (hipGraph-capturable).
//
// Network (models/arch/Unet.py:6-91): 5 scales, channels 32/64/128/256/512, 18 conv3x3+LeakyReLU,
// 4 maxpool2, 4 transposed conv 2x2/s2, 4 channel concats [up, skip] (never materialised: the conv
// kernels read two sources), 1x1 head.
#include <mutex>
#include <unordered_map>
#include <math.h>
#include "unet_misc.h"

namespace {

constexpr int NLEV = 5;
inline int chan(int l) { return 32 << l; }

struct LayerDef {
    int kind;        // 0 conv3x3, 1 convT2x2, 2 head 1x1
    int cin, cout;
    size_t w_off, b_off;      // flat parameter offsets (floats)
};

// layer order == named_parameters() order of the reference module
enum {
    L_E0A = 0, L_E0B, L_E1A, L_E1B, L_E2A, L_E2B, L_E3A, L_E3B, L_E4A, L_E4B,
    L_UP3, L_D3A, L_D3B, L_UP2, L_D2A, L_D2B, L_UP1, L_D1A, L_D1B, L_UP0, L_D0A, L_D0B, L_HEAD, NLAYERS
};

size_t build_layers(int in_ch, int out_ch, LayerDef* L) {
    int k = 0;
    for (int l = 0; l < NLEV; ++l) {
        L[k++] = {0, l == 0 ? in_ch : chan(l - 1), chan(l), 0, 0};
        L[k++] = {0, chan(l), chan(l), 0, 0};
    }
    for (int l = 3; l >= 0; --l) {
        L[k++] = {1, chan(l + 1), chan(l), 0, 0};
        L[k++] = {0, 2 * chan(l), chan(l), 0, 0};
        L[k++] = {0, chan(l), chan(l), 0, 0};
    }
    L[k++] = {2, 32, out_ch, 0, 0};
    size_t off = 0;
    for (int i = 0; i < NLAYERS; ++i) {
        const size_t taps = L[i].kind == 0 ? 9 : (L[i].kind == 1 ? 4 : 1);
        L[i].w_off = off; off += (size_t)L[i].cin * L[i].cout * taps;
        L[i].b_off = off; off += L[i].cout;
    }
    return off;
}

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

int choose_psplit(int groups, int ntiles) {
    int ps = 512 / groups;
    if (ps < 1) ps = 1;
    if (ps > ntiles) ps = ntiles;
    if (ps < 1) ps = 1;
    return ps;
}

struct WgradGeom { int T, CA, CBp, groups, ntiles, psplit, w8; size_t floats; };

// algo: the fp32 product scheme of the call (the three-piece scheme re-blocks 3x3 layers for wgrad8_kernel: one 8-wave workgroup per CU)
WgradGeom wgrad_geom(int mode, int CA, int CB, int N, int H, int W, int algo) {
    WgradGeom g;
    g.T = mode == CONV_3X3 ? 9 : 4;
    g.CA = CA;
    g.CBp = (CB + 31) / 32 * 32;
    g.w8 = 0;
    int cob8, jb8, th8, tw8;
    // algo 3 = bf16 tensors on the same re-blocked kernel (one plane per operand, 256-pixel tiles)
    if ((algo == 1 || algo == 3) && mode == CONV_3X3 && CB % 32 == 0 && wgrad8_shape(CA, g.CBp, cob8, jb8, th8, tw8, algo == 3)) {
        g.w8 = 1;
        g.groups = (CA / cob8) * (g.CBp / jb8);
        g.ntiles = wgrad8_ntiles(CA, g.CBp, N, H, W, algo == 3);
        int ps = 256 / g.groups;
        if (ps > g.ntiles) ps = g.ntiles;
        g.psplit = ps < 1 ? 1 : ps;
        g.floats = (size_t)g.psplit * ((size_t)g.T * CA * g.CBp + CA);
        return g;
    }
    // algo 1, transposed convs with >= 128 x 64 channels: wgradt8_kernel (one 8-wave workgroup per CU, 128 x 64 blocks, 8 x 8-pixel tiles per image)
    if (algo == 1 && mode == CONV_GATHER2X2 && CB == g.CBp && wgradt8_takes(CA, CB)) {
        g.w8 = 1;
        g.groups = (CA / 128) * (g.CBp / 64);
        g.ntiles = wgradt8_ntiles(N, H, W);
        int ps = 256 / g.groups;
        if (ps > g.ntiles) ps = g.ntiles;
        g.psplit = ps < 1 ? 1 : ps;
        g.floats = (size_t)g.psplit * ((size_t)g.T * CA * g.CBp + CA);
        return g;
    }
    const int COB = (CA % 64 == 0) ? 64 : 32;
    const int TH = mode == CONV_3X3 ? 4 : 2;
    g.groups = (CA / COB) * (g.CBp / 32);
    g.ntiles = ((W + 31) / 32) * ((H + TH - 1) / TH) * N;
    g.psplit = choose_psplit(g.groups, g.ntiles);
    g.floats = (size_t)g.psplit * ((size_t)g.T * CA * g.CBp + CA);
    return g;
}

// ------------------------------------------------------------------------------------------------
// layer helpers on packed weights
// ------------------------------------------------------------------------------------------------
// Operand bounds for the two-piece fp16 product scheme (conv_fp32_algo 2): set by the caller right before a helper call,
// consumed (and cleared) by it.  Slots are device floats in the workspace.
struct Amax { const float* in0 = nullptr; const float* in1 = nullptr; const float* w = nullptr; float* out0 = nullptr; float* out1 = nullptr; };
thread_local Amax g_am;
// scratch for the split-K partial sums of small problems (conv_x3.hip): the plan's weight-gradient partial region, free while a conv runs;
// set for the duration of a U-Net entry point
struct KPart { float* p = nullptr; size_t floats = 0; };
thread_local KPart g_kp;
struct KPartScope { KPart prev; KPartScope(float* p, size_t n) : prev(g_kp) { g_kp.p = p; g_kp.floats = n; } ~KPartScope() { g_kp = prev; } };
thread_local int g_algo = -1;      // fp32 product scheme of the entry point being executed on this thread (-1: process default)
struct AlgoScope { int prev; explicit AlgoScope(int a) : prev(g_algo) { g_algo = resolve_algo(a); } ~AlgoScope() { g_algo = prev; } };
inline void take_amax(ConvArgs& a) { a.algo = g_algo; a.amax_in0 = g_am.in0; a.amax_in1 = g_am.in1; a.amax_w = g_am.w; a.amax_out0 = g_am.out0; a.amax_out1 = g_am.out1; g_am = Amax(); }
inline void take_amax(WgradArgs& a) { a.algo = g_algo; a.amax_g = g_am.in0; a.amax_x0 = g_am.w; a.amax_x1 = g_am.in1; g_am = Amax(); }      // wgrad: in0 = G, w = X0, in1 = X1
enum { S_W = 0, S_X = 23, S_EA = 24, S_EB = 29, S_UP = 34, S_DA = 38, S_DB = 42, S_GA = 46, S_GB = 47, S_SKIP = 48, S_COUNT = 64 };
int conv_fwd(const float* in0, int C0, const float* in1, int C1, const float* wp, const float* bias, float* out, int N, int H, int W,
             int Cout, int lrelu, hipStream_t st, float* pool_out = nullptr, unsigned* codes_out = nullptr, unsigned* pool_codes_out = nullptr) {
    ConvArgs a = {};
    a.pool_out = pool_out;
    a.codes_out = codes_out;
    a.pool_codes_out = pool_codes_out;
    a.in0 = in0; a.in1 = in1; a.C0 = C0; a.C1 = C1; a.wp = wp; a.N = N; a.H = H; a.W = W; a.Nout = Cout;
    a.epi = EPI_FWD; a.bias = bias; a.lrelu = lrelu; a.out0 = out;
    a.kpart = g_kp.p; a.kpart_floats = g_kp.floats;
    take_amax(a);
    return launch_conv(a, CONV_3X3, st);
}

// g: [N,H,W,Cout] -> din (Cin channels, split over out0/out1), wb packed [9][Cin][Cout]
int conv_bwd_data(const float* g, const float* wb, float* out0, float* out1, int split, const float* act0, const float* act1, int N, int H,
                  int W, int Cin, int Cout, hipStream_t st, const unsigned* codes0 = nullptr) {
    ConvArgs a = {};
    a.codes0 = codes0;
    a.in0 = g; a.C0 = Cout; a.wp = wb; a.N = N; a.H = H; a.W = W; a.Nout = Cin;
    a.epi = EPI_GRAD; a.out0 = out0; a.out1 = out1; a.split = split; a.act0 = act0; a.act1 = act1;
    a.kpart = g_kp.p; a.kpart_floats = g_kp.floats;
    take_amax(a);
    return launch_conv(a, CONV_3X3, st);
}

int conv_wgrad(const float* g, int Cout, const float* x0, int C0, const float* x1, int C1, int Cin_real, float* dw, float* db, float* part,
               int N, int H, int W, hipStream_t st) {
    const WgradGeom q = wgrad_geom(CONV_3X3, Cout, C0 + C1, N, H, W, (C0 % 32 || C1 % 32) ? 0 : g_algo);
    WgradArgs a = {};
    a.g = g; a.CA = Cout; a.x0 = x0; a.x1 = x1; a.C0 = C0; a.C1 = C1; a.N = N; a.H = H; a.W = W;
    a.part = part; a.bpart = db ? part + (size_t)q.psplit * q.T * q.CA * q.CBp : nullptr; a.CBp = q.CBp; a.psplit = q.psplit;
    take_amax(a);
    a.wgrad8 = q.w8;
    int rc = launch_wgrad(a, CONV_3X3, st);
    if (rc) return rc;
    return launch_wgrad_reduce(part, a.bpart, dw, db, q.psplit, q.T, q.CA, q.CBp, Cin_real, st);
}

// in [N,H,W,Cin] -> out [N,2H,2W,Cout]; wf packed [4*Cout][Cin]
int convt_fwd(const float* in, const float* wf, const float* bias, float* out, int N, int H, int W, int Cin, int Cout, hipStream_t st) {
    ConvArgs a = {};
    a.in0 = in; a.C0 = Cin; a.wp = wf; a.N = N; a.H = H; a.W = W; a.Nout = 4 * Cout;
    a.epi = EPI_CONVT_FWD; a.bias = bias; a.out0 = out; a.Cout_t = Cout;
    take_amax(a);
    return launch_conv(a, CONV_1X1, st);
}

// dout [N,2H,2W,Cout] -> din [N,H,W,Cin] (times slope(act) if act); wb packed [4][Cin][Cout]
int convt_bwd_data(const float* dout, const float* wb, const float* act, float* din, int N, int H, int W, int Cin, int Cout, hipStream_t st) {
    ConvArgs a = {};
    a.in0 = dout; a.C0 = Cout; a.wp = wb; a.N = N; a.H = H; a.W = W; a.Nout = Cin;
    a.epi = EPI_GRAD; a.out0 = din; a.split = Cin; a.act0 = act;
    take_amax(a);
    return launch_conv(a, CONV_GATHER2X2, st);
}

// dw[ci][co][tap] = sum in[p][ci] * dout[gather(p,tap)][co];  db[co] = column sums of dout
int convt_wgrad(const float* in, const float* dout, float* dw, float* db, float* part, int N, int H, int W, int Cin, int Cout, hipStream_t st) {
    const WgradGeom q = wgrad_geom(CONV_GATHER2X2, Cin, Cout, N, H, W, g_algo == 1 ? 1 : 0);
    WgradArgs a = {};
    a.g = in; a.CA = Cin; a.x0 = dout; a.C0 = Cout; a.N = N; a.H = H; a.W = W;
    a.part = part; a.bpart = nullptr; a.CBp = q.CBp; a.psplit = q.psplit; a.wgrad8 = q.w8;
    // bias gradient = column sums of dout: accumulated by the weight-gradient kernel from its staging registers (every dout pixel
    // passes through exactly once per block column); the partials take the plan's bias slot (psplit * CA floats >= psplit * CBp)
    const bool fused_bias = db != nullptr && q.CBp <= q.CA && Cout == q.CBp;
    a.xbpart = fused_bias ? part + (size_t)q.psplit * q.T * q.CA * q.CBp : nullptr;
    take_amax(a);
    int rc = launch_wgrad(a, CONV_GATHER2X2, st);
    if (rc) return rc;
    // (the bias partials ride in the same reduction launch: one launch less per transposed conv)
    rc = launch_wgrad_reduce(part, fused_bias ? a.xbpart : nullptr, dw, fused_bias ? db : nullptr, q.psplit, q.T, q.CA, q.CBp, Cout, st, q.CBp);
    if (rc || !db || fused_bias) return rc;
    return launch_colsum(dout, db, part, (size_t)N * 4 * H * W, Cout, st);
}

// ------------------------------------------------------------------------------------------------
// workspace plan
// ------------------------------------------------------------------------------------------------
struct Plan {
    LayerDef L[NLAYERS];
    size_t nparams;
    int N, H, W, in_ch, out_ch;
    int Hl[NLEV], Wl[NLEV];
    // offsets in floats
    size_t wp_fwd[NLAYERS], wp_bwd[NLAYERS];
    size_t x16, ea[NLEV], eb[NLEV], pool[NLEV - 1], up[NLEV - 1], da[NLEV - 1], db[NLEV - 1];
    size_t gA, gB, skip[NLEV - 1], part, part_floats, amax;
    size_t head_part; // partials of the fused training head (eld_unet_forward_loss_ex -> the backward)
    // slope codes (conv.h ConvArgs::codes_out) of the activations the 32- / 64-channel backward-data epilogues multiply by: ea[0], ea[1], da[0], da[1]
    // (2 bits per element); written by the fp32 three-piece forward where `codes` is set (see make_plan) and, for ea[0] / da[0], by the bf16 forward
    // (unet_forward_bf16); read by the backward for the regions the last forward reports (FusedFwd::codes)
    size_t cd_ea[2], cd_da[2];
    // round 6: slope codes of eb[0], eb[1] and argmax codes of pool[0], pool[1] (ConvArgs::pool_codes_out): what maxpool_bwd_codes_kernel reads instead of eb
    size_t cd_eb[2], pc[2];
    bool codes;
    size_t total;     // floats
};

int make_plan(Plan& P, int N, int H, int W, int in_ch, int out_ch) {
    if (N < 1 || H < 16 || W < 16 || (H % 16) || (W % 16)) return ELD_EINVAL;     // 4 pool levels (Unet.py:51-63)
    if (in_ch < 1 || in_ch > 16 || out_ch < 1 || out_ch > 4) return ELD_ENOTSUP;
    P.N = N; P.H = H; P.W = W; P.in_ch = in_ch; P.out_ch = out_ch;
    P.nparams = build_layers(in_ch, out_ch, P.L);
    for (int l = 0; l < NLEV; ++l) { P.Hl[l] = H >> l; P.Wl[l] = W >> l; }
    size_t off = 0;
    auto take = [&](size_t n) { const size_t o = off; off += align_up(n, 64); return o; };   // 256-byte aligned
    for (int i = 0; i < NLAYERS; ++i) {
        const LayerDef& d = P.L[i];
        if (d.kind == 0) {
            const int cinp = (d.cin + 15) / 16 * 16;
            P.wp_fwd[i] = take((size_t)9 * d.cout * cinp * 2);          // fp32 [tap][n][c] or the pre-split slab layout (7 B per element)
            P.wp_bwd[i] = (i == L_E0A) ? 0 : take((size_t)9 * d.cin * d.cout * 2);
        } else if (d.kind == 1) {
            P.wp_fwd[i] = take((size_t)4 * d.cout * d.cin);
            P.wp_bwd[i] = take((size_t)4 * d.cin * d.cout);
        } else {
            P.wp_fwd[i] = P.wp_bwd[i] = 0;
        }
    }
    auto act = [&](int l, int c) { return (size_t)N * P.Hl[l] * P.Wl[l] * c; };
    P.x16 = take(act(0, 16));
    for (int l = 0; l < NLEV; ++l) { P.ea[l] = take(act(l, chan(l))); P.eb[l] = take(act(l, chan(l))); }
    for (int l = 0; l < NLEV - 1; ++l) {
        P.pool[l] = take(act(l + 1, chan(l)));
        P.up[l] = take(act(l, chan(l)));
        P.da[l] = take(act(l, chan(l)));
        P.db[l] = take(act(l, chan(l)));
        P.skip[l] = take(act(l, chan(l)));
    }
    P.gA = take(act(0, 32));
    P.gB = take(act(0, 32));
    P.head_part = take(head_train_ws_floats());
    size_t pmax = head_bwd_ws_floats();
    pmax = pmax > conv_first_wgrad_ws_floats() ? pmax : conv_first_wgrad_ws_floats();
    pmax = pmax > colsum_ws_floats(256) ? pmax : colsum_ws_floats(256);
    for (int i = 0; i < NLAYERS; ++i) {
        const LayerDef& d = P.L[i];
        int lev;
        if (i <= L_E4B) lev = i / 2; else if (i < L_HEAD) lev = 3 - (i - L_UP3) / 3; else lev = 0;
        size_t f = 0;
        if (d.kind == 0) {
            f = wgrad_geom(CONV_3X3, d.cout, i == L_E0A ? 16 : d.cin, N, P.Hl[lev], P.Wl[lev], 0).floats;
            const size_t f1 = wgrad_geom(CONV_3X3, d.cout, i == L_E0A ? 16 : d.cin, N, P.Hl[lev], P.Wl[lev], 1).floats;
            f = f1 > f ? f1 : f;
            const size_t f3 = wgrad_geom(CONV_3X3, d.cout, i == L_E0A ? 16 : d.cin, N, P.Hl[lev], P.Wl[lev], 3).floats;
            f = f3 > f ? f3 : f;
        } else if (d.kind == 1) {
            f = wgrad_geom(CONV_GATHER2X2, d.cin, d.cout, N, P.Hl[lev + 1], P.Wl[lev + 1], 0).floats;
            const size_t f1 = wgrad_geom(CONV_GATHER2X2, d.cin, d.cout, N, P.Hl[lev + 1], P.Wl[lev + 1], 1).floats;
            f = f1 > f ? f1 : f;
        }
        pmax = f > pmax ? f : pmax;
    }
    // split-K partial sums of small problems share this region: up to 16 parts of the largest 3x3 output that can take the split
    for (int i = 0; i < NLAYERS; ++i) {
        const LayerDef& d = P.L[i];
        if (d.kind != 0) continue;
        int lev;
        if (i <= L_E4B) lev = i / 2; else lev = 3 - (i - L_UP3) / 3;
        const long long px8 = (long long)((P.Wl[lev] + 31) / 32) * ((P.Hl[lev] + 7) / 8) * N;
        for (int dir = 0; dir < 2; ++dir) {
            const int nout = dir ? d.cin : d.cout, k = dir ? d.cout : d.cin;
            if (nout % 64 || k % 32) continue;
            long long ks = eld_num_cus() / (px8 * (nout / 64));
            if (ks > k / 32) ks = k / 32;
            if (ks > 16) ks = 16;
            if (ks >= 2) { const size_t f = (size_t)ks * N * P.Hl[lev] * P.Wl[lev] * nout; pmax = f > pmax ? f : pmax; }
        }
    }
    P.part = take(pmax);
    P.part_floats = pmax;
    // Slope codes only where the 64-channel level-1 layers run the unsplit 8-wave kernel (conv_x3.hip x3_slab_bn: every CU gets a 16-row tile): the
    // split-K finish kernel of small problems does not write them.  Which regions a forward actually filled travels to the backward through the
    // workspace's CodesState (the fp32 scheme, the first layer's kernel and the debug mask are per-call choices).
    P.codes = conv_tile_count(N, P.Hl[1], P.Wl[1], 16, false) >= eld_num_cus();
    for (int l = 0; l < 2; ++l) {
        P.cd_ea[l] = take(act(l, chan(l)) / 16 + 64);
        P.cd_da[l] = take(act(l, chan(l)) / 16 + 64);
        P.cd_eb[l] = take(act(l, chan(l)) / 16 + 64);
        P.pc[l] = take(act(l + 1, chan(l)) / 16 + 64);
    }
    P.amax = take(S_COUNT);
    P.total = off;
    return 0;
}

// dir: PACK_FWD / PACK_BWD = the layouts one direction reads; PACK_BOTH = both in ONE launch (the fused training forward: the backward that follows
// with dout == NULL then skips its own pack launch -- the parameters it differentiates are by construction the ones the forward ran with)
enum { PACK_FWD = 0, PACK_BWD = 1, PACK_BOTH = 2 };
int pack_weights(const Plan& P, const float* params, float* ws, int dir, hipStream_t st, bool bf16 = false, float* amax = nullptr) {
    PackJobs jobs;
    jobs.n = 0;
    for (int pass = 0; pass < 2; ++pass) {
    const bool for_backward = pass == 1;
    if ((dir == PACK_FWD && for_backward) || (dir == PACK_BWD && !for_backward)) continue;
    for (int i = 0; i < NLAYERS; ++i) {
        const LayerDef& d = P.L[i];
        PackJob J = {};
        J.src_off = d.w_off; J.Cout = d.cout; J.Cin = d.cin; J.Cinp = d.cin;
        if (d.kind == 0) {
            J.T = 9;
            if (!for_backward) { J.kind = PACK_CONV_FWD; J.Cinp = (d.cin + 15) / 16 * 16; J.dst_off = P.wp_fwd[i]; if (bf16 && i == L_E0A) continue; }
            else if (i != L_E0A) { J.kind = PACK_CONV_BWD; J.dst_off = P.wp_bwd[i]; }
            else continue;
        } else if (d.kind == 1) {
            J.T = 4;
            J.kind = for_backward ? PACK_CONVT_BWD : PACK_CONVT_FWD;
            J.dst_off = for_backward ? P.wp_bwd[i] : P.wp_fwd[i];
        } else {
            continue;
        }
        J.bf16 = bf16 ? 1 : 0;
        J.amax_slot = S_W + i;
        // three-piece scheme: 3x3 layers with GEMM N % 64 == 0 get pre-split slabs (N = Cout forward, Cin backward-data)
        int lev;                                   // resolution level the layer runs at (its tile domain decides the slab's channel-block width)
        if (i <= L_E4B) lev = i / 2; else if (i < L_HEAD) lev = 3 - (i - L_UP3) / 3; else lev = 0;
        J.x3bn = (!bf16 && g_algo == 1 && d.kind == 0) ? x3_slab_bn(for_backward ? d.cin : d.cout, P.N, P.Hl[lev], P.Wl[lev]) : 0;
        // bf16: 3x3 layers the DMA kernel takes get its slab layout (GEMM N = Cout forward / Cin backward-data, K the other one)
        J.bfdbn = (bf16 && d.kind == 0) ? (for_backward ? bfd_slab_bn(d.cin, d.cout, P.N, P.Hl[lev], P.Wl[lev]) : bfd_slab_bn(d.cout, J.Cinp, P.N, P.Hl[lev], P.Wl[lev])) : 0;
        // bf16 transposed convs on conv_bfg_kernel take its slab layout (forward: GEMM N = 4 Cout, K = Cin on the input-resolution domain of level lev+1;
        // backward-data: N = Cin, K = 4 Cout)
        J.bfgbn = (bf16 && d.kind == 1) ? (for_backward ? bfg_slab_bn(true, d.cin, d.cout, 0, P.N, P.Hl[lev + 1], P.Wl[lev + 1])
                                                        : bfg_slab_bn(false, 4 * d.cout, d.cin, d.cout, P.N, P.Hl[lev + 1], P.Wl[lev + 1])) : 0;
        jobs.job[jobs.n++] = J;
    }
    }
    return launch_pack_all(jobs, params, ws, st, amax);
}

#define RC(x) do { int rc__ = (x); if (rc__) return rc__; } while (0)

// Gradient buckets for the data-parallel exchange: the backward produces parameter gradients in DESCENDING flat-buffer order
// (head, decoder level 0 .. 3, encoder level 4 .. 0 -- the reverse of named_parameters()), so once layer i's weight/bias
// gradients are enqueued every gradient at offset >= L[i].w_off is final.  done(i) records the events of all buckets that
// start at or above that offset; the caller's communication stream waits on them and all-reduces bucket by bucket while the
// rest of the backward runs.
struct BucketMarks {
    const int64_t* start = nullptr;     // ascending parameter offsets (floats)
    void* const* event = nullptr;       // hipEvent_t per bucket
    int n = 0, next = 0;                // buckets [next .. n) counted from the TOP are not recorded yet
    int done(const Plan& P, int layer, hipStream_t st) {
        while (n - 1 - next >= 0 && start[n - 1 - next] >= (int64_t)P.L[layer].w_off) {
            hipError_t e = hipEventRecord((hipEvent_t)event[n - 1 - next], st);
            if (e != hipSuccess) return (int)e;
            ++next;
        }
        return 0;
    }
};

// hl != null: the training step's fused head (output + loss + head backward), see launch_head_train
struct HeadLoss { const float* target; float* loss; int mse; float grad_scale; };

// What eld_unet_forward_loss_ex leaves for the backward that follows it with dout == NULL (besides the head's partials): both pack directions done, and
// -- when the first layer reads the NCHW input directly -- the caller's input tensor itself instead of a copy in the workspace (8 frames: 388 MB, 0.13 ms).
struct FusedFwd { bool packed = false; const float* x = nullptr; unsigned codes = 0; };
// FusedFwd::codes / the forwards' have_out: which slope-code regions the LAST forward on this workspace filled (it decides per launch: kernel choice,
// fp32 scheme, debug mask) -- the backward reads a region only when its bit is set, whatever its own switches say
enum { CODES_EA0 = 1, CODES_EA1 = 2, CODES_DA0 = 4, CODES_DA1 = 8, CODES_EB0 = 16, CODES_EB1 = 32, CODES_INFER = 256 };      // CODES_EBl: slope codes of eb[l] AND argmax codes of pool[l]      // CODES_INFER: the forward was eld_unet_infer_ex -- nothing for a backward

int unet_forward(const Plan& P, const float* x, const float* prm, float* out, float* ws, hipStream_t st, const HeadLoss* hl = nullptr, unsigned* have_out = nullptr,
                 bool infer = false) {
    const int N = P.N;
    KPartScope kp(ws + P.part, P.part_floats);
    const bool h2 = g_algo == 2;                                 // operand bounds ride along in the workspace
    float* am = ws + P.amax;
    auto AM = [&](int in0, int in1, int w, int out0) {
        if (!h2) return;
        g_am.in0 = am + in0; g_am.in1 = in1 >= 0 ? am + in1 : nullptr; g_am.w = am + w; g_am.out0 = am + out0;
    };
    if (h2 && hipMemsetAsync(am, 0, S_COUNT * sizeof(float), st) != hipSuccess) return (int)hipGetLastError();
    RC(pack_weights(P, prm, ws, hl ? PACK_BOTH : PACK_FWD, st, false, h2 ? am : nullptr));
    const bool first_direct = P.in_ch <= 4;        // conv1_1 straight from the NCHW planes (conv_first.hip)
    const bool use_codes = !infer && P.codes && g_algo == 1 && !(debug_kernel_mask(-1) & 128);   // slope codes for the backward-data epilogues of levels 0 / 1 (Plan::codes)
    auto CD = [&](size_t off) -> unsigned* { return use_codes ? reinterpret_cast<unsigned*>(ws + off) : nullptr; };
    const bool use_pcodes = use_codes && !(debug_kernel_mask(-1) & 256);      // argmax codes of the two full-size pools (bit 8 of the mask: A/B and bit-identity switch)
    if (have_out) *have_out = !use_codes ? 0u : ((!first_direct || conv_first_writes_codes(P.in_ch)) ? CODES_EA0 : 0u) | CODES_EA1 | CODES_DA0 | CODES_DA1 | (use_pcodes ? (x3w_enabled() ? CODES_EB0 : 0u) | CODES_EB1 : 0u);
    if (first_direct && (hl || infer)) {
        // fused training forward: the backward reads the caller's x (include/eld_amd.h: it must stay valid and unchanged until that call);
        // inference (eld_unet_infer_ex): no backward follows
    } else if (first_direct) {
        // keep the input for the backward's weight gradient (the backward entry point does not receive x)
        hipError_t e = hipMemcpyAsync(ws + P.x16, x, (size_t)N * P.in_ch * P.H * P.W * sizeof(float), hipMemcpyDeviceToDevice, st);
        if (e != hipSuccess) return (int)e;
    } else {
        RC(launch_nchw_to_nhwc16(x, ws + P.x16, N, P.in_ch, P.H, P.W, st));
    }
    for (int l = 0; l < NLEV; ++l) {
        const LayerDef& A = P.L[2 * l]; const LayerDef& B = P.L[2 * l + 1];
        const float* src = l == 0 ? ws + P.x16 : ws + P.pool[l - 1];
        const int cin = l == 0 ? 16 : chan(l - 1);
        if (l == 0 && first_direct) {
            RC(launch_conv_first_fwd(x, prm + A.w_off, prm + A.b_off, ws + P.ea[0], N, P.in_ch, P.H, P.W, 1, st, CD(P.cd_ea[0])));
            if (h2) RC(launch_absmax(ws + P.ea[0], (size_t)N * P.H * P.W * chan(0), am + S_EA, st));
        } else {
            if (h2 && l == 0) RC(launch_absmax(ws + P.x16, (size_t)N * P.H * P.W * 16, am + S_X, st));
            AM(l == 0 ? S_X : S_EB + l - 1, -1, S_W + 2 * l, S_EA + l);          // pooled input is bounded by its source's bound
            RC(conv_fwd(src, cin, nullptr, 0, ws + P.wp_fwd[2 * l], prm + A.b_off, ws + P.ea[l], N, P.Hl[l], P.Wl[l], chan(l), 1, st, nullptr,
                        l < 2 ? CD(P.cd_ea[l]) : nullptr));
        }
        AM(S_EA + l, -1, S_W + 2 * l + 1, S_EB + l);
        // three-piece scheme: the conv's epilogue also writes the pooled tensor (no second pass over eb[l])
        const bool fuse_pool = g_algo == 1 && l < NLEV - 1;
        const bool pcodes = fuse_pool && l < 2 && use_pcodes && (l == 1 || x3w_enabled());      // level 0: conv_x3w_kernel writes them, conv_x3_kernel<32, 4> does not       // levels 0 / 1: the pool's backward then reads codes instead of eb[l]
        RC(conv_fwd(ws + P.ea[l], chan(l), nullptr, 0, ws + P.wp_fwd[2 * l + 1], prm + B.b_off, ws + P.eb[l], N, P.Hl[l], P.Wl[l], chan(l), 1, st,
                    fuse_pool ? ws + P.pool[l] : nullptr, pcodes ? CD(P.cd_eb[l]) : nullptr, pcodes ? CD(P.pc[l]) : nullptr));
        if (l < NLEV - 1 && !fuse_pool) RC(launch_maxpool_fwd(ws + P.eb[l], ws + P.pool[l], N, P.Hl[l + 1], P.Wl[l + 1], chan(l), st));
    }
    for (int l = 3; l >= 0; --l) {
        const int iu = L_UP3 + 3 * (3 - l);
        const float* src = l == 3 ? ws + P.eb[4] : ws + P.db[l + 1];
        AM(l == 3 ? S_EB + 4 : S_DB + l + 1, -1, S_W + iu, S_UP + l);
        RC(convt_fwd(src, ws + P.wp_fwd[iu], prm + P.L[iu].b_off, ws + P.up[l], N, P.Hl[l + 1], P.Wl[l + 1], chan(l + 1), chan(l), st));
        AM(S_UP + l, S_EB + l, S_W + iu + 1, S_DA + l);
        RC(conv_fwd(ws + P.up[l], chan(l), ws + P.eb[l], chan(l), ws + P.wp_fwd[iu + 1], prm + P.L[iu + 1].b_off, ws + P.da[l], N, P.Hl[l], P.Wl[l], chan(l), 1, st, nullptr,
                    l < 2 ? CD(P.cd_da[l]) : nullptr));
        AM(S_DA + l, -1, S_W + iu + 2, S_DB + l);
        RC(conv_fwd(ws + P.da[l], chan(l), nullptr, 0, ws + P.wp_fwd[iu + 2], prm + P.L[iu + 2].b_off, ws + P.db[l], N, P.Hl[l], P.Wl[l], chan(l), 1, st));
    }
    const LayerDef& Hd = P.L[L_HEAD];
    if (hl)      // training step: output, loss, the head's backward and the gradient of conv9_2's output (-> gA) in one pass
        return launch_head_train(ws + P.db[0], 0, prm + Hd.w_off, prm + Hd.b_off, hl->target, out, ws + P.gA, ws + P.head_part, hl->loss, N, P.H, P.W, P.out_ch,
                                 hl->mse, hl->grad_scale, st);
    RC(launch_head_fwd(ws + P.db[0], prm + Hd.w_off, prm + Hd.b_off, out, N, P.H, P.W, P.out_ch, st));
    return 0;
}

// bf16 forward (inference): bf16 activations and packed weights, fp32 accumulation / bias; the first layer reads the
// fp32 NCHW planes and the head writes fp32 NCHW.  Buffers of the fp32 plan are reused (half filled).
int conv_fwd_bf16(const bf16_t* in0, int C0, const bf16_t* in1, int C1, const bf16_t* wp, const float* bias, bf16_t* out, int N, int H, int W,
                  int Cout, int lrelu, hipStream_t st, bf16_t* pool_out = nullptr, unsigned* codes_out = nullptr) {
    ConvArgs a = {};
    a.pool_out = pool_out;
    a.codes_out = codes_out;
    a.in0 = in0; a.in1 = in1; a.C0 = C0; a.C1 = C1; a.wp = wp; a.N = N; a.H = H; a.W = W; a.Nout = Cout;
    a.epi = EPI_FWD; a.bias = bias; a.lrelu = lrelu; a.out0 = out; a.dtype = DT_BF16;
    return launch_conv(a, CONV_3X3, st);
}

int unet_forward_bf16(const Plan& P, const float* x, const float* prm, float* out, float* ws, hipStream_t st, const HeadLoss* hl = nullptr, unsigned* have_out = nullptr,
                      bool infer = false) {
    const int N = P.N;
    if (P.in_ch > 4) return ELD_ENOTSUP;
    RC(pack_weights(P, prm, ws, hl ? PACK_BOTH : PACK_FWD, st, true));
    if (!hl && !infer) {   // keep the fp32 input for the first layer's weight gradient (the fused training forward leaves it with the caller: see FusedFwd)
        hipError_t e = hipMemcpyAsync(ws + P.x16, x, (size_t)N * P.in_ch * P.H * P.W * sizeof(float), hipMemcpyDeviceToDevice, st);
        if (e != hipSuccess) return (int)e;
    }
    auto B = [&](size_t off) { return reinterpret_cast<bf16_t*>(ws + off); };
    // Slope codes of the two full-resolution activations whose backward-data epilogues run on conv_bfs_kernel (conv9_2's and conv1_2's: HBM-bound
    // launches that would otherwise re-read the saved 32-channel tensor): written by conv1_1's and conv9_1's epilogues where those kernels are the
    // ones that run (conv_first's MFMA kernel; conv_bfs for the 64 -> 32 layer) -- 8 bytes per pixel instead of 64 in the backward
    const bool bfs0 = !infer && bfs_takes(32, 32, N, P.H, P.W) && !(debug_kernel_mask(-1) & 128);
    const bool c_ea0 = bfs0 && conv_first_writes_codes(P.in_ch), c_da0 = bfs0 && bfs_takes(32, 64, N, P.H, P.W);
    if (have_out) *have_out = (c_ea0 ? CODES_EA0 : 0u) | (c_da0 ? CODES_DA0 : 0u);
    for (int l = 0; l < NLEV; ++l) {
        const LayerDef& A = P.L[2 * l]; const LayerDef& Bd = P.L[2 * l + 1];
        if (l == 0)
            RC(launch_conv_first_fwd_bf16(x, prm + A.w_off, prm + A.b_off, B(P.ea[0]), N, P.in_ch, P.H, P.W, 1, st,
                                          c_ea0 ? reinterpret_cast<unsigned*>(ws + P.cd_ea[0]) : nullptr));
        else
            RC(conv_fwd_bf16(B(P.pool[l - 1]), chan(l - 1), nullptr, 0, B(P.wp_fwd[2 * l]), prm + A.b_off, B(P.ea[l]), N, P.Hl[l], P.Wl[l], chan(l), 1, st));
        // layers on the DMA kernel (conv_bfd.hip) write the pooled tensor from their epilogue
        const bool fuse_pool = l < NLEV - 1 && bfd_slab_bn(chan(l), chan(l), N, P.Hl[l], P.Wl[l]) != 0;
        RC(conv_fwd_bf16(B(P.ea[l]), chan(l), nullptr, 0, B(P.wp_fwd[2 * l + 1]), prm + Bd.b_off, B(P.eb[l]), N, P.Hl[l], P.Wl[l], chan(l), 1, st,
                         fuse_pool ? B(P.pool[l]) : nullptr));
        if (l < NLEV - 1 && !fuse_pool) RC(launch_maxpool_fwd_bf16(B(P.eb[l]), B(P.pool[l]), N, P.Hl[l + 1], P.Wl[l + 1], chan(l), st));
    }
    for (int l = 3; l >= 0; --l) {
        const int iu = L_UP3 + 3 * (3 - l);
        const bf16_t* src = l == 3 ? B(P.eb[4]) : B(P.db[l + 1]);
        ConvArgs a = {};
        a.in0 = src; a.C0 = chan(l + 1); a.wp = B(P.wp_fwd[iu]); a.N = N; a.H = P.Hl[l + 1]; a.W = P.Wl[l + 1]; a.Nout = 4 * chan(l);
        a.epi = EPI_CONVT_FWD; a.bias = prm + P.L[iu].b_off; a.out0 = B(P.up[l]); a.Cout_t = chan(l); a.dtype = DT_BF16;
        RC(launch_conv(a, CONV_1X1, st));
        RC(conv_fwd_bf16(B(P.up[l]), chan(l), B(P.eb[l]), chan(l), B(P.wp_fwd[iu + 1]), prm + P.L[iu + 1].b_off, B(P.da[l]), N, P.Hl[l], P.Wl[l], chan(l), 1, st, nullptr,
                         l == 0 && c_da0 ? reinterpret_cast<unsigned*>(ws + P.cd_da[0]) : nullptr));
        RC(conv_fwd_bf16(B(P.da[l]), chan(l), nullptr, 0, B(P.wp_fwd[iu + 2]), prm + P.L[iu + 2].b_off, B(P.db[l]), N, P.Hl[l], P.Wl[l], chan(l), 1, st));
    }
    const LayerDef& Hd = P.L[L_HEAD];
    if (hl)
        return launch_head_train(B(P.db[0]), 1, prm + Hd.w_off, prm + Hd.b_off, hl->target, out, B(P.gA), ws + P.head_part, hl->loss, N, P.H, P.W, P.out_ch,
                                 hl->mse, hl->grad_scale, st);
    RC(launch_head_fwd_bf16(B(P.db[0]), prm + Hd.w_off, prm + Hd.b_off, out, N, P.H, P.W, P.out_ch, st));
    return 0;
}

int unet_backward(const Plan& P, const float* dout, const float* prm, float* grd, float* ws, hipStream_t st, BucketMarks& marks, const FusedFwd& fused) {
    const int N = P.N;
    KPartScope kp(ws + P.part, P.part_floats);
    const bool h2 = g_algo == 2;
    float* am = ws + P.amax;
    if (h2 && hipMemsetAsync(am + S_GA, 0, (S_COUNT - S_GA) * sizeof(float), st) != hipSuccess) return (int)hipGetLastError();
    if (!fused.packed) RC(pack_weights(P, prm, ws, PACK_BWD, st, false, h2 ? am : nullptr));
    float* gA = ws + P.gA; float* gB = ws + P.gB; float* part = ws + P.part;
    // slope codes the forward left for levels 0 / 1 (FusedFwd::codes names the regions it filled)
    auto CD = [&](size_t off, unsigned bit) -> const unsigned* { return g_algo == 1 && (fused.codes & bit) ? reinterpret_cast<const unsigned*>(ws + off) : nullptr; };
    auto gs = [&](const float* buf) { return buf == gA ? (int)S_GA : (int)S_GB; };                 // slot of a ping-pong gradient buffer
    auto fresh = [&](int slot) -> int {                                                               // zero a slot before its tensor is rewritten
        if (!h2) return 0;
        return hipMemsetAsync(am + slot, 0, sizeof(float), st) == hipSuccess ? 0 : (int)hipGetLastError();
    };
    auto WG = [&](int g, int x0, int x1) { if (h2) { g_am.in0 = am + g; g_am.w = am + x0; g_am.in1 = x1 >= 0 ? am + x1 : nullptr; } };
    auto BD = [&](int g, int w, int o0, int o1) { if (h2) { g_am.in0 = am + g; g_am.w = am + w; g_am.out0 = am + o0; g_am.out1 = o1 >= 0 ? am + o1 : nullptr; } };
    const LayerDef& Hd = P.L[L_HEAD];
    // head: g (pre-activation grad of conv9_2) -> gA
    if (dout) RC(launch_head_bwd(dout, ws + P.db[0], prm + Hd.w_off, gA, grd + Hd.w_off, grd + Hd.b_off, part, N, P.H, P.W, P.out_ch, st));
    else RC(launch_head_train_reduce(ws + P.head_part, grd + Hd.w_off, grd + Hd.b_off, N, P.H, P.W, P.out_ch, 0, st));      // gA and the partials: eld_unet_forward_loss_ex
    RC(marks.done(P, L_HEAD, st));
    if (h2) RC(launch_absmax(gA, (size_t)N * P.H * P.W * 32, am + S_GA, st));
    float* cur = gA; float* oth = gB;
    for (int l = 0; l <= 3; ++l) {            // decoder levels 0 (conv9) .. 3 (conv6)
        const int iu = L_UP3 + 3 * (3 - l);
        const int H = P.Hl[l], W = P.Wl[l], C = chan(l);
        // conv_2 of the level: input da[l]
        WG(gs(cur), S_DA + l, -1);
        RC(conv_wgrad(cur, C, ws + P.da[l], C, nullptr, 0, C, grd + P.L[iu + 2].w_off, grd + P.L[iu + 2].b_off, part, N, H, W, st));
        RC(marks.done(P, iu + 2, st));
        RC(fresh(gs(oth))); BD(gs(cur), S_W + iu + 2, gs(oth), -1);
        RC(conv_bwd_data(cur, ws + P.wp_bwd[iu + 2], oth, nullptr, C, ws + P.da[l], nullptr, N, H, W, C, C, st, l < 2 ? CD(P.cd_da[l], l ? CODES_DA1 : CODES_DA0) : nullptr));
        { float* t = cur; cur = oth; oth = t; }
        // conv_1: input cat[up[l], eb[l]] -> d_up (raw) in oth, skip grad (raw) in skip[l]
        WG(gs(cur), S_UP + l, S_EB + l);
        RC(conv_wgrad(cur, C, ws + P.up[l], C, ws + P.eb[l], C, 2 * C, grd + P.L[iu + 1].w_off, grd + P.L[iu + 1].b_off, part, N, H, W, st));
        RC(marks.done(P, iu + 1, st));
        RC(fresh(gs(oth))); BD(gs(cur), S_W + iu + 1, gs(oth), S_SKIP + l);
        RC(conv_bwd_data(cur, ws + P.wp_bwd[iu + 1], oth, ws + P.skip[l], C, nullptr, nullptr, N, H, W, 2 * C, C, st));
        { float* t = cur; cur = oth; oth = t; }
        // transposed conv: input src (level l+1, 2C channels), output grad = cur (d_up)
        const float* src = l == 3 ? ws + P.eb[4] : ws + P.db[l + 1];
        WG(l == 3 ? S_EB + 4 : S_DB + l + 1, gs(cur), -1);
        RC(convt_wgrad(src, cur, grd + P.L[iu].w_off, grd + P.L[iu].b_off, part, N, P.Hl[l + 1], P.Wl[l + 1], 2 * C, C, st));
        RC(marks.done(P, iu, st));
        RC(fresh(gs(oth))); BD(gs(cur), S_W + iu, gs(oth), -1);
        RC(convt_bwd_data(cur, ws + P.wp_bwd[iu], src, oth, N, P.Hl[l + 1], P.Wl[l + 1], 2 * C, C, st));
        { float* t = cur; cur = oth; oth = t; }
    }
    // cur = pre-activation grad of conv5_2 (level 4)
    for (int l = 4; l >= 0; --l) {
        const int H = P.Hl[l], W = P.Wl[l], C = chan(l);
        const int ia = 2 * l, ib = 2 * l + 1;
        WG(gs(cur), S_EA + l, -1);
        RC(conv_wgrad(cur, C, ws + P.ea[l], C, nullptr, 0, C, grd + P.L[ib].w_off, grd + P.L[ib].b_off, part, N, H, W, st));
        RC(marks.done(P, ib, st));
        RC(fresh(gs(oth))); BD(gs(cur), S_W + ib, gs(oth), -1);
        RC(conv_bwd_data(cur, ws + P.wp_bwd[ib], oth, nullptr, C, ws + P.ea[l], nullptr, N, H, W, C, C, st,
                         l < 2 ? CD(P.cd_ea[l], l ? CODES_EA1 : CODES_EA0) : nullptr));
        { float* t = cur; cur = oth; oth = t; }
        if (l == 0) {
            if (P.in_ch <= 4)
                RC(launch_conv_first_wgrad(cur, fused.x ? fused.x : ws + P.x16, grd + P.L[ia].w_off, grd + P.L[ia].b_off, part, N, P.in_ch, H, W, st, g_algo == 1));
            else {
                WG(gs(cur), S_X, -1);
                RC(conv_wgrad(cur, C, ws + P.x16, 16, nullptr, 0, P.in_ch, grd + P.L[ia].w_off, grd + P.L[ia].b_off, part, N, H, W, st));
            }
            RC(marks.done(P, ia, st));
            break;
        }
        const int Cp = chan(l - 1);
        WG(gs(cur), S_EB + l - 1, -1);                 // pooled activations are bounded by their source's bound
        RC(conv_wgrad(cur, C, ws + P.pool[l - 1], Cp, nullptr, 0, Cp, grd + P.L[ia].w_off, grd + P.L[ia].b_off, part, N, H, W, st));
        RC(marks.done(P, ia, st));
        RC(fresh(gs(oth))); BD(gs(cur), S_W + ia, gs(oth), -1);
        RC(conv_bwd_data(cur, ws + P.wp_bwd[ia], oth, nullptr, Cp, nullptr, nullptr, N, H, W, Cp, C, st));      // d_pool (raw)
        { float* t = cur; cur = oth; oth = t; }
        if (l - 1 < 2 && CD(P.cd_eb[l - 1], l - 1 ? CODES_EB1 : CODES_EB0) != nullptr)
            RC(launch_maxpool_bwd_codes(CD(P.pc[l - 1], l - 1 ? CODES_EB1 : CODES_EB0), CD(P.cd_eb[l - 1], l - 1 ? CODES_EB1 : CODES_EB0), cur, ws + P.skip[l - 1], oth, N, H, W, Cp, st));
        else
            RC(launch_maxpool_bwd(ws + P.eb[l - 1], cur, ws + P.skip[l - 1], oth, N, H, W, Cp, st));
        if (h2) { RC(fresh(gs(oth))); RC(launch_absmax(oth, (size_t)N * 4 * H * W * Cp, am + gs(oth), st)); }
        { float* t = cur; cur = oth; oth = t; }
    }
    return 0;
}

// bf16 backward: activation gradients bf16 (igemm on v_mfma_f32_32x32x16_bf16), weight gradients accumulated and written
// in fp32 (the wgrad kernels convert the bf16 operands while staging; fp32 MFMA).  Mirrors unet_backward step for step.
int conv_bwd_data_bf16(const bf16_t* g, const bf16_t* wb, bf16_t* out0, bf16_t* out1, int split, const bf16_t* act0, const bf16_t* act1, int N, int H,
                       int W, int Cin, int Cout, hipStream_t st, const unsigned* codes0 = nullptr) {
    ConvArgs a = {};
    a.codes0 = codes0;
    a.in0 = g; a.C0 = Cout; a.wp = wb; a.N = N; a.H = H; a.W = W; a.Nout = Cin;
    a.epi = EPI_GRAD; a.out0 = out0; a.out1 = out1; a.split = split; a.act0 = act0; a.act1 = act1; a.dtype = DT_BF16;
    return launch_conv(a, CONV_3X3, st);
}

int conv_wgrad_bf16(const bf16_t* g, int Cout, const bf16_t* x0, int C0, const bf16_t* x1, int C1, float* dw, float* db, float* part,
                    int N, int H, int W, hipStream_t st) {
    const WgradGeom q = wgrad_geom(CONV_3X3, Cout, C0 + C1, N, H, W, (C0 % 32 || C1 % 32) ? 0 : 3);
    WgradArgs a = {};
    a.g = g; a.CA = Cout; a.x0 = x0; a.x1 = x1; a.C0 = C0; a.C1 = C1; a.N = N; a.H = H; a.W = W; a.dtype = DT_BF16;
    a.part = part; a.bpart = db ? part + (size_t)q.psplit * q.T * q.CA * q.CBp : nullptr; a.CBp = q.CBp; a.psplit = q.psplit;
    a.wgrad8 = q.w8;
    RC(launch_wgrad(a, CONV_3X3, st));
    return launch_wgrad_reduce(part, a.bpart, dw, db, q.psplit, q.T, q.CA, q.CBp, C0 + C1, st);
}

int unet_backward_bf16(const Plan& P, const float* dout, const float* prm, float* grd, float* ws, hipStream_t st, BucketMarks& marks, const FusedFwd& fused) {
    const int N = P.N;
    if (P.in_ch > 4) return ELD_ENOTSUP;
    if (!fused.packed) RC(pack_weights(P, prm, ws, PACK_BWD, st, true));
    auto B = [&](size_t off) { return reinterpret_cast<bf16_t*>(ws + off); };
    bf16_t* cur = B(P.gA); bf16_t* oth = B(P.gB); float* part = ws + P.part;
    const LayerDef& Hd = P.L[L_HEAD];
    if (dout) RC(launch_head_bwd_bf16(dout, B(P.db[0]), prm + Hd.w_off, cur, grd + Hd.w_off, grd + Hd.b_off, part, N, P.H, P.W, P.out_ch, st));
    else RC(launch_head_train_reduce(ws + P.head_part, grd + Hd.w_off, grd + Hd.b_off, N, P.H, P.W, P.out_ch, 1, st));
    RC(marks.done(P, L_HEAD, st));
    auto swap = [&]() { bf16_t* t = cur; cur = oth; oth = t; };
    for (int l = 0; l <= 3; ++l) {
        const int iu = L_UP3 + 3 * (3 - l);
        const int H = P.Hl[l], W = P.Wl[l], C = chan(l);
        RC(conv_wgrad_bf16(cur, C, B(P.da[l]), C, nullptr, 0, grd + P.L[iu + 2].w_off, grd + P.L[iu + 2].b_off, part, N, H, W, st));
        RC(marks.done(P, iu + 2, st));
        RC(conv_bwd_data_bf16(cur, B(P.wp_bwd[iu + 2]), oth, nullptr, C, B(P.da[l]), nullptr, N, H, W, C, C, st,
                              l == 0 && (fused.codes & CODES_DA0) ? reinterpret_cast<const unsigned*>(ws + P.cd_da[0]) : nullptr));
        swap();
        RC(conv_wgrad_bf16(cur, C, B(P.up[l]), C, B(P.eb[l]), C, grd + P.L[iu + 1].w_off, grd + P.L[iu + 1].b_off, part, N, H, W, st));
        RC(marks.done(P, iu + 1, st));
        RC(conv_bwd_data_bf16(cur, B(P.wp_bwd[iu + 1]), oth, B(P.skip[l]), C, nullptr, nullptr, N, H, W, 2 * C, C, st));
        swap();
        const bf16_t* src = l == 3 ? B(P.eb[4]) : B(P.db[l + 1]);
        {   // transposed conv: weight gradient (gather mode), bias gradient (column sums of d_up), backward data
            const int Hi = P.Hl[l + 1], Wi = P.Wl[l + 1];
            const WgradGeom q = wgrad_geom(CONV_GATHER2X2, 2 * C, C, N, Hi, Wi, 0);
            WgradArgs a = {};
            a.g = src; a.CA = 2 * C; a.x0 = cur; a.C0 = C; a.N = N; a.H = Hi; a.W = Wi; a.dtype = DT_BF16;
            a.part = part; a.bpart = nullptr; a.CBp = q.CBp; a.psplit = q.psplit;
            a.xbpart = part + (size_t)q.psplit * q.T * q.CA * q.CBp;          // bias gradient = column sums of d_up, from the staging registers
            RC(launch_wgrad(a, CONV_GATHER2X2, st));
            RC(launch_wgrad_reduce(part, a.xbpart, grd + P.L[iu].w_off, grd + P.L[iu].b_off, q.psplit, q.T, q.CA, q.CBp, C, st, q.CBp));
            RC(marks.done(P, iu, st));
            ConvArgs c = {};
            c.in0 = cur; c.C0 = C; c.wp = B(P.wp_bwd[iu]); c.N = N; c.H = Hi; c.W = Wi; c.Nout = 2 * C;
            c.epi = EPI_GRAD; c.out0 = oth; c.split = 2 * C; c.act0 = src; c.dtype = DT_BF16;
            RC(launch_conv(c, CONV_GATHER2X2, st));
        }
        swap();
    }
    for (int l = 4; l >= 0; --l) {
        const int H = P.Hl[l], W = P.Wl[l], C = chan(l);
        const int ia = 2 * l, ib = 2 * l + 1;
        RC(conv_wgrad_bf16(cur, C, B(P.ea[l]), C, nullptr, 0, grd + P.L[ib].w_off, grd + P.L[ib].b_off, part, N, H, W, st));
        RC(marks.done(P, ib, st));
        RC(conv_bwd_data_bf16(cur, B(P.wp_bwd[ib]), oth, nullptr, C, B(P.ea[l]), nullptr, N, H, W, C, C, st,
                              l == 0 && (fused.codes & CODES_EA0) ? reinterpret_cast<const unsigned*>(ws + P.cd_ea[0]) : nullptr));
        swap();
        if (l == 0) {
            RC(launch_conv_first_wgrad_bf16(cur, fused.x ? fused.x : ws + P.x16, grd + P.L[ia].w_off, grd + P.L[ia].b_off, part, N, P.in_ch, H, W, st));
            RC(marks.done(P, ia, st));
            break;
        }
        const int Cp = chan(l - 1);
        RC(conv_wgrad_bf16(cur, C, B(P.pool[l - 1]), Cp, nullptr, 0, grd + P.L[ia].w_off, grd + P.L[ia].b_off, part, N, H, W, st));
        RC(marks.done(P, ia, st));
        RC(conv_bwd_data_bf16(cur, B(P.wp_bwd[ia]), oth, nullptr, Cp, nullptr, nullptr, N, H, W, Cp, C, st));
        swap();
        RC(launch_maxpool_bwd_bf16(B(P.eb[l - 1]), cur, B(P.skip[l - 1]), oth, N, H, W, Cp, st));
        swap();
    }
    return 0;
}

}  // namespace

// ====================================================================================================
// C ABI
// ====================================================================================================
extern "C" int eld_unet_param_offsets(int in_ch, int out_ch, int64_t* offsets) {
    if (!offsets || in_ch < 1 || in_ch > 16 || out_ch < 1 || out_ch > 4) return ELD_EINVAL;
    LayerDef L[NLAYERS];
    const size_t n = build_layers(in_ch, out_ch, L);
    for (int i = 0; i < NLAYERS; ++i) { offsets[2 * i] = (int64_t)L[i].w_off; offsets[2 * i + 1] = (int64_t)L[i].b_off; }
    offsets[ELD_UNET_NTENSORS] = (int64_t)n;
    return 0;
}

extern "C" size_t eld_unet_workspace_bytes(int N, int H, int W, int in_ch, int out_ch) {
    Plan P;
    if (make_plan(P, N, H, W, in_ch, out_ch)) return 0;
    return P.total * sizeof(float);
}

// Which forward last filled a workspace (host call order = stream order for calls on one stream): eld_unet_backward_ex with dout == NULL consumes
// the gradient buffer and the head partials that ONLY eld_unet_forward_loss_ex leaves there, for exactly its N / H / W / precision.  Keyed by the
// workspace pointer; a plain forward on the same workspace clears the entry.  Host bookkeeping only: no device read, no synchronisation.
namespace {
struct HeadState { int N, H, W, in_ch, out_ch, precision; const float* x; };
// Which slope-code regions the last forward on a workspace filled, and for which problem (FusedFwd::codes): same bookkeeping, for EVERY forward
struct CodesState { int N, H, W, in_ch, out_ch, precision; unsigned have; };
// One table for both, keyed by the workspace pointer.  Workspaces come and go with their owners, so the table is bounded -- by evicting the ONE
// least recently touched entry, never by clearing it: an entry that is still between its forward and its backward is by construction among the most
// recently touched ones (it would take WS_STATE_MAX other workspaces run in between to push it out).  What a lost entry costs: no codes entry sends
// the backward to the saved activations (same bits); no head entry makes the dout == NULL backward fail with ELD_EINVAL.
struct WsState { HeadState head; bool has_head = false; CodesState codes; bool has_codes = false; unsigned long long stamp = 0; };
constexpr size_t WS_STATE_MAX = 256;
std::mutex g_head_mu;
std::unordered_map<const void*, WsState> g_ws_state;
unsigned long long g_ws_clock = 0;
WsState& ws_state_touch(const void* ws) {          // caller holds g_head_mu
    auto it = g_ws_state.find(ws);
    if (it == g_ws_state.end()) {
        if (g_ws_state.size() >= WS_STATE_MAX) {
            auto old = g_ws_state.begin();
            for (auto j = g_ws_state.begin(); j != g_ws_state.end(); ++j) if (j->second.stamp < old->second.stamp) old = j;
            g_ws_state.erase(old);
        }
        it = g_ws_state.emplace(ws, WsState()).first;
    }
    it->second.stamp = ++g_ws_clock;
    return it->second;
}
void ws_state_drop_if_empty(const void* ws) {       // caller holds g_head_mu
    const auto it = g_ws_state.find(ws);
    if (it != g_ws_state.end() && !it->second.has_head && !it->second.has_codes) g_ws_state.erase(it);
}
void codes_state_set(const void* ws, const CodesState& cs) {
    std::lock_guard<std::mutex> lk(g_head_mu);
    if (cs.have == 0) {
        const auto it = g_ws_state.find(ws);
        if (it != g_ws_state.end()) { it->second.has_codes = false; ws_state_drop_if_empty(ws); }
        return;
    }
    WsState& w = ws_state_touch(ws);
    w.codes = cs; w.has_codes = true;
}
unsigned codes_state_get(const void* ws, int N, int H, int W, int in_ch, int out_ch, int precision) {
    std::lock_guard<std::mutex> lk(g_head_mu);
    const auto it = g_ws_state.find(ws);
    if (it == g_ws_state.end() || !it->second.has_codes) return 0;
    it->second.stamp = ++g_ws_clock;
    const CodesState& c = it->second.codes;
    return c.N == N && c.H == H && c.W == W && c.in_ch == in_ch && c.out_ch == out_ch && c.precision == precision ? c.have : 0u;
}
void head_state_set(const void* ws, const HeadState* st) {
    std::lock_guard<std::mutex> lk(g_head_mu);
    if (st) {
        WsState& w = ws_state_touch(ws);
        w.head = *st; w.has_head = true;
    } else {
        const auto it = g_ws_state.find(ws);
        if (it != g_ws_state.end()) { it->second.has_head = false; ws_state_drop_if_empty(ws); }
    }
}
bool head_state_is(const void* ws, HeadState& want) {      // fills want.x (the fused forward's input tensor) on a match
    std::lock_guard<std::mutex> lk(g_head_mu);
    const auto it = g_ws_state.find(ws);
    if (it == g_ws_state.end() || !it->second.has_head) return false;
    it->second.stamp = ++g_ws_clock;
    const HeadState& h = it->second.head;
    if (!(h.N == want.N && h.H == want.H && h.W == want.W && h.in_ch == want.in_ch && h.out_ch == want.out_ch && h.precision == want.precision)) return false;
    want.x = h.x;
    return true;
}
}  // namespace

static int unet_entry_checks(Plan& P, const void* a, const void* b, const void* c, const void* ws, size_t ws_bytes, int N, int H, int W, int in_ch, int out_ch) {
    RC(make_plan(P, N, H, W, in_ch, out_ch));
    if (!a || !b || !c || !ws) return ELD_EINVAL;
    if (ws_bytes < P.total * sizeof(float)) return ELD_EWS;
    return 0;
}

static int unet_forward_entry(const float* x, const float* params, float* out, void* ws, size_t ws_bytes, int N, int H, int W,
                              int in_ch, int out_ch, int precision, int fp32_algo, void* stream, bool infer) {
    if (N == 0) return 0;
    if ((precision != 0 && precision != 1) || fp32_algo > 2) return ELD_EINVAL;
    Plan P;
    RC(unet_entry_checks(P, x, params, out, ws, ws_bytes, N, H, W, in_ch, out_ch));
    AlgoScope scope(fp32_algo);
    head_state_set(ws, nullptr);      // whatever fused head state the workspace held is overwritten
    unsigned have = 0;
    const int rc = precision == 1 ? unet_forward_bf16(P, x, params, out, (float*)ws, as_stream(stream), nullptr, &have, infer)
                                  : unet_forward(P, x, params, out, (float*)ws, as_stream(stream), nullptr, &have, infer);
    codes_state_set(ws, CodesState{N, H, W, in_ch, out_ch, precision, rc == 0 ? (infer ? (unsigned)CODES_INFER : have) : 0u});
    return rc;
}
extern "C" int eld_unet_forward_ex(const float* x, const float* params, float* out, void* ws, size_t ws_bytes, int N, int H, int W,
                                   int in_ch, int out_ch, int precision, int fp32_algo, void* stream) {
    return unet_forward_entry(x, params, out, ws, ws_bytes, N, H, W, in_ch, out_ch, precision, fp32_algo, stream, false);
}
extern "C" int eld_unet_infer_ex(const float* x, const float* params, float* out, void* ws, size_t ws_bytes, int N, int H, int W,
                                 int in_ch, int out_ch, int precision, int fp32_algo, void* stream) {
    return unet_forward_entry(x, params, out, ws, ws_bytes, N, H, W, in_ch, out_ch, precision, fp32_algo, stream, true);
}

extern "C" int eld_unet_forward_loss_ex(const float* x, const float* params, const float* target, float* out, float* loss, void* ws, size_t ws_bytes,
                                        int N, int H, int W, int in_ch, int out_ch, int precision, int fp32_algo, int loss_kind, float grad_scale, void* stream) {
    if (N == 0) return 0;
    if ((precision != 0 && precision != 1) || fp32_algo > 2 || (loss_kind != 0 && loss_kind != 1) || !target || !loss) return ELD_EINVAL;
    Plan P;
    RC(unet_entry_checks(P, x, params, out, ws, ws_bytes, N, H, W, in_ch, out_ch));
    AlgoScope scope(fp32_algo);
    const HeadLoss hl = {target, loss, loss_kind, grad_scale};
    head_state_set(ws, nullptr);
    unsigned have = 0;
    const int rc = precision == 1 ? unet_forward_bf16(P, x, params, out, (float*)ws, as_stream(stream), &hl, &have)
                                  : unet_forward(P, x, params, out, (float*)ws, as_stream(stream), &hl, &have);
    codes_state_set(ws, CodesState{N, H, W, in_ch, out_ch, precision, rc == 0 ? have : 0u});
    if (rc == 0) { const HeadState hs = {N, H, W, in_ch, out_ch, precision, x}; head_state_set(ws, &hs); }
    return rc;
}
extern "C" int eld_unet_backward_ex(const float* dout, const float* params, float* grads, void* ws, size_t ws_bytes, int N, int H, int W,
                                    int in_ch, int out_ch, int precision, int fp32_algo, const int64_t* bucket_start, void* const* bucket_event,
                                    int n_buckets, void* stream) {
    if (N == 0) return 0;
    if ((precision != 0 && precision != 1) || fp32_algo > 2) return ELD_EINVAL;
    Plan P;
    RC(unet_entry_checks(P, dout ? (const void*)dout : (const void*)params, params, grads, ws, ws_bytes, N, H, W, in_ch, out_ch));      // dout == NULL: the head's share was done by eld_unet_forward_loss_ex
    if (n_buckets < 0 || (n_buckets > 0 && (!bucket_start || !bucket_event))) return ELD_EINVAL;
    // dout == NULL is only meaningful right after eld_unet_forward_loss_ex on this workspace with the same problem: anything else would hand back
    // stale head gradients without a sign of trouble
    FusedFwd fused;
    {
        HeadState hs = {N, H, W, in_ch, out_ch, precision, nullptr};
        const bool after_fused = head_state_is(ws, hs);
        if (!dout && !after_fused) return ELD_EINVAL;
        // An explicit dout after the fused-loss forward is a valid call order too (the head is then recomputed from dout): that forward left the
        // input with the caller instead of copying it into the workspace, so the first layer's weight gradient must read it there as well.
        if (after_fused) fused.x = in_ch <= 4 ? hs.x : nullptr;      // (more input planes: the forward converted x to NHWC16 in the workspace)
        fused.packed = !dout;                                         // with an explicit dout the backward packs its own weights (params may have changed)
        fused.codes = codes_state_get(ws, N, H, W, in_ch, out_ch, precision);
        if (fused.codes & CODES_INFER) return ELD_EINVAL;               // the last forward on this workspace was eld_unet_infer_ex: it kept nothing for a backward
    }
    for (int k = 0; k < n_buckets; ++k)
        if (!bucket_event[k] || bucket_start[k] < 0 || (k > 0 && bucket_start[k] <= bucket_start[k - 1]) || (size_t)bucket_start[k] >= P.nparams) return ELD_EINVAL;
    BucketMarks marks;
    marks.start = bucket_start; marks.event = bucket_event; marks.n = n_buckets;
    AlgoScope scope(fp32_algo);
    const int rc = precision == 1 ? unet_backward_bf16(P, dout, params, grads, (float*)ws, as_stream(stream), marks, fused)
                                  : unet_backward(P, dout, params, grads, (float*)ws, as_stream(stream), marks, fused);
    if (rc) return rc;
    return marks.next == n_buckets ? 0 : ELD_EINVAL;
}

// the round-1 entry points: the fp32 product scheme is the process default (eld_conv_fp32_algo) at the time of EACH call
extern "C" int eld_unet_forward(const float* x, const float* params, float* out, void* ws, size_t ws_bytes, int N, int H, int W,
                                int in_ch, int out_ch, void* stream) {
    return eld_unet_forward_ex(x, params, out, ws, ws_bytes, N, H, W, in_ch, out_ch, 0, -1, stream);
}
extern "C" int eld_unet_forward_bf16(const float* x, const float* params, float* out, void* ws, size_t ws_bytes, int N, int H, int W,
                                     int in_ch, int out_ch, void* stream) {
    return eld_unet_forward_ex(x, params, out, ws, ws_bytes, N, H, W, in_ch, out_ch, 1, -1, stream);
}
extern "C" int eld_unet_backward(const float* dout, const float* params, float* grads, void* ws, size_t ws_bytes, int N, int H, int W,
                                 int in_ch, int out_ch, void* stream) {
    return eld_unet_backward_ex(dout, params, grads, ws, ws_bytes, N, H, W, in_ch, out_ch, 0, -1, nullptr, nullptr, 0, stream);
}
extern "C" int eld_unet_backward_bf16(const float* dout, const float* params, float* grads, void* ws, size_t ws_bytes, int N, int H, int W,
                                      int in_ch, int out_ch, void* stream) {
    return eld_unet_backward_ex(dout, params, grads, ws, ws_bytes, N, H, W, in_ch, out_ch, 1, -1, nullptr, nullptr, 0, stream);
}
extern "C" int eld_unet_backward_buckets(const float* dout, const float* params, float* grads, void* ws, size_t ws_bytes, int N, int H, int W,
                                         int in_ch, int out_ch, int precision, const int64_t* bucket_start, void* const* bucket_event,
                                         int n_buckets, void* stream) {
    return eld_unet_backward_ex(dout, params, grads, ws, ws_bytes, N, H, W, in_ch, out_ch, precision, -1, bucket_start, bucket_event, n_buckets, stream);
}

extern "C" int eld_debug_ws_state_entries(void) { std::lock_guard<std::mutex> lk(g_head_mu); return (int)g_ws_state.size(); }

extern "C" void eld_debug_conv_prof(void* buf) { conv_x3_set_prof((unsigned long long*)buf); }

extern "C" int eld_conv_fp32_algo(int algo) { return conv_fp32_algo(algo); }

extern "C" size_t eld_l1_workspace_bytes(void) { return l1_ws_floats() * sizeof(float); }

extern "C" int eld_l1_loss(const float* out, const float* target, float* dout, float* loss, void* ws, size_t n, float grad_scale, void* stream) {
    if (!out || !target || !loss || !ws || n == 0) return ELD_EINVAL;
    return launch_l1(out, target, dout, loss, (float*)ws, n, grad_scale, as_stream(stream));
}

extern "C" int eld_mse_loss(const float* out, const float* target, float* dout, float* loss, void* ws, size_t n, float grad_scale, void* stream) {
    if (!out || !target || !loss || !ws || n == 0) return ELD_EINVAL;
    return launch_mse(out, target, dout, loss, (float*)ws, n, grad_scale, as_stream(stream));
}

extern "C" int eld_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, size_t n, double lr, double beta1,
                             double beta2, double eps, double weight_decay, int step, double grad_scale, void* stream) {
    if (n == 0) return 0;
    if (!params || !grads || !exp_avg || !exp_avg_sq || step < 1) return ELD_EINVAL;
    return launch_adam(params, grads, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, step, grad_scale, as_stream(stream));
}

// ---- single layers -----------------------------------------------------------------------------------
extern "C" size_t eld_layer_workspace_bytes(int N, int H, int W, int Cin, int Cout) {
    if (N < 1 || H < 1 || W < 1 || Cin < 1 || Cout < 1) return 0;
    const int cinp = (Cin + 15) / 16 * 16;
    size_t f = (size_t)9 * Cout * cinp * 2 + 64;                            // one packed weight set (fp32 or pre-split slabs)
    size_t p = 0, q;
    if (Cout % 32 == 0) { q = wgrad_geom(CONV_3X3, Cout, Cin, N, H, W, 0).floats; p = q > p ? q : p; q = wgrad_geom(CONV_3X3, Cout, Cin, N, H, W, 1).floats; p = q > p ? q : p; }
    if (Cin % 32 == 0) { q = wgrad_geom(CONV_GATHER2X2, Cin, Cout, N, H, W, 0).floats; p = q > p ? q : p; q = wgrad_geom(CONV_GATHER2X2, Cin, Cout, N, H, W, 1).floats; p = q > p ? q : p; }
    q = colsum_ws_floats(Cout > Cin ? Cout : Cin); p = q > p ? q : p;
    return (align_up(f, 64) + align_up(p, 64) + 64) * sizeof(float);      // + 64 operand-bound slots (conv_fp32_algo 2)
}

// operand bounds of a single-layer call under conv_fp32_algo 2: slots 0..2 = inputs (a, b, c), 3..4 = outputs
static int layer_amax(void* ws, size_t ws_bytes, hipStream_t st, const float* a, size_t na, const float* b, size_t nb, const float* c, size_t nc, float** slots) {
    *slots = nullptr;
    if (g_algo != 2) return 0;
    float* am = (float*)ws + ws_bytes / sizeof(float) - 64;
    if (hipMemsetAsync(am, 0, 64 * sizeof(float), st) != hipSuccess) return (int)hipGetLastError();
    if (a) RC(launch_absmax(a, na, am + 0, st));
    if (b) RC(launch_absmax(b, nb, am + 1, st));
    if (c) RC(launch_absmax(c, nc, am + 2, st));
    *slots = am;
    return 0;
}

static int layer_ws(void* ws, size_t ws_bytes, int N, int H, int W, int Cin, int Cout, float** pack, float** part) {
    const size_t need = eld_layer_workspace_bytes(N, H, W, Cin, Cout);
    if (!ws || need == 0) return ELD_EINVAL;
    if (ws_bytes < need) return ELD_EWS;
    const int cinp = (Cin + 15) / 16 * 16;
    *pack = (float*)ws;
    *part = (float*)ws + align_up((size_t)9 * Cout * cinp * 2 + 64, 64);
    return 0;
}

extern "C" int eld_conv3x3_forward(const float* in0, int C0, const float* in1, int C1, const float* w, const float* bias, float* out, int N,
                                   int H, int W, int Cout, int lrelu, void* ws, size_t ws_bytes, void* stream) {
    if (N == 0) return 0;
    AlgoScope scope(-1);
    if (!in0 || !w || !bias || !out || (C0 + C1) % 16 || C0 % 16 || Cout % 32) return ELD_EINVAL;
    float *pack, *part;
    RC(layer_ws(ws, ws_bytes, N, H, W, C0 + C1, Cout, &pack, &part));
    hipStream_t st = as_stream(stream);
    RC(launch_pack(w, pack, PACK_CONV_FWD, Cout, C0 + C1, C0 + C1, 9, st, g_algo == 1 ? x3_slab_bn(Cout, N, H, W) : 0));
    float* am;
    RC(layer_amax(ws, eld_layer_workspace_bytes(N, H, W, C0 + C1, Cout), st, in0, (size_t)N * H * W * C0, in1, (size_t)N * H * W * C1, pack, (size_t)9 * Cout * (C0 + C1), &am));
    if (am) { g_am.in0 = am; g_am.in1 = in1 ? am + 1 : nullptr; g_am.w = am + 2; g_am.out0 = am + 3; }
    return conv_fwd(in0, C0, in1, C1, pack, bias, out, N, H, W, Cout, lrelu, st);
}

extern "C" int eld_conv3x3_backward_data(const float* g, const float* w, float* din0, float* din1, int split, const float* act0, const float* act1,
                                         int N, int H, int W, int Cin, int Cout, void* ws, size_t ws_bytes, void* stream) {
    if (N == 0) return 0;
    AlgoScope scope(-1);
    if (!g || !w || !din0 || Cin % 32 || Cout % 16 || split % 32 || split > Cin || (split < Cin && !din1)) return ELD_EINVAL;
    float *pack, *part;
    RC(layer_ws(ws, ws_bytes, N, H, W, Cin, Cout, &pack, &part));
    hipStream_t st = as_stream(stream);
    RC(launch_pack(w, pack, PACK_CONV_BWD, Cout, Cin, Cin, 9, st, g_algo == 1 ? x3_slab_bn(Cin, N, H, W) : 0));
    float* am;
    RC(layer_amax(ws, eld_layer_workspace_bytes(N, H, W, Cin, Cout), st, g, (size_t)N * H * W * Cout, nullptr, 0, pack, (size_t)9 * Cout * Cin, &am));
    if (am) { g_am.in0 = am; g_am.w = am + 2; g_am.out0 = am + 3; g_am.out1 = am + 4; }
    return conv_bwd_data(g, pack, din0, din1, split, act0, act1, N, H, W, Cin, Cout, st);
}

extern "C" int eld_conv3x3_backward_weight(const float* g, const float* x0, int C0, const float* x1, int C1, float* dw, float* db, int N, int H,
                                           int W, int Cout, void* ws, size_t ws_bytes, void* stream) {
    if (N == 0) return 0;
    AlgoScope scope(-1);
    if (!g || !x0 || !dw || Cout % 32 || C0 % 4 || C1 % 4) return ELD_EINVAL;
    float *pack, *part;
    RC(layer_ws(ws, ws_bytes, N, H, W, C0 + C1, Cout, &pack, &part));
    float* am;
    RC(layer_amax(ws, eld_layer_workspace_bytes(N, H, W, C0 + C1, Cout), as_stream(stream), g, (size_t)N * H * W * Cout, x1, (size_t)N * H * W * C1, x0, (size_t)N * H * W * C0, &am));
    if (am) { g_am.in0 = am; g_am.in1 = x1 ? am + 1 : nullptr; g_am.w = am + 2; }
    return conv_wgrad(g, Cout, x0, C0, x1, C1, C0 + C1, dw, db, part, N, H, W, as_stream(stream));
}

extern "C" int eld_convt2x2_forward(const float* in, const float* w, const float* bias, float* out, int N, int H, int W, int Cin, int Cout,
                                    void* ws, size_t ws_bytes, void* stream) {
    if (N == 0) return 0;
    AlgoScope scope(-1);
    if (!in || !w || !bias || !out || Cin % 16 || Cout % 8) return ELD_EINVAL;
    float *pack, *part;
    RC(layer_ws(ws, ws_bytes, N, H, W, Cin, Cout, &pack, &part));
    hipStream_t st = as_stream(stream);
    RC(launch_pack(w, pack, PACK_CONVT_FWD, Cout, Cin, Cin, 4, st));
    float* am;
    RC(layer_amax(ws, eld_layer_workspace_bytes(N, H, W, Cin, Cout), st, in, (size_t)N * H * W * Cin, nullptr, 0, pack, (size_t)4 * Cout * Cin, &am));
    if (am) { g_am.in0 = am; g_am.w = am + 2; g_am.out0 = am + 3; }
    return convt_fwd(in, pack, bias, out, N, H, W, Cin, Cout, st);
}

extern "C" int eld_convt2x2_backward_data(const float* dout, const float* w, const float* act, float* din, int N, int H, int W, int Cin, int Cout,
                                          void* ws, size_t ws_bytes, void* stream) {
    if (N == 0) return 0;
    AlgoScope scope(-1);
    if (!dout || !w || !din || Cin % 32 || Cout % 16) return ELD_EINVAL;
    float *pack, *part;
    RC(layer_ws(ws, ws_bytes, N, H, W, Cin, Cout, &pack, &part));
    hipStream_t st = as_stream(stream);
    RC(launch_pack(w, pack, PACK_CONVT_BWD, Cout, Cin, Cin, 4, st));
    float* am;
    RC(layer_amax(ws, eld_layer_workspace_bytes(N, H, W, Cin, Cout), st, dout, (size_t)N * 4 * H * W * Cout, nullptr, 0, pack, (size_t)4 * Cout * Cin, &am));
    if (am) { g_am.in0 = am; g_am.w = am + 2; g_am.out0 = am + 3; }
    return convt_bwd_data(dout, pack, act, din, N, H, W, Cin, Cout, st);
}

extern "C" int eld_convt2x2_backward_weight(const float* in, const float* dout, float* dw, float* db, int N, int H, int W, int Cin, int Cout,
                                            void* ws, size_t ws_bytes, void* stream) {
    if (N == 0) return 0;
    AlgoScope scope(-1);
    if (!in || !dout || !dw || Cin % 32 || Cout % 4) return ELD_EINVAL;
    float *pack, *part;
    RC(layer_ws(ws, ws_bytes, N, H, W, Cin, Cout, &pack, &part));
    float* am;
    RC(layer_amax(ws, eld_layer_workspace_bytes(N, H, W, Cin, Cout), as_stream(stream), in, (size_t)N * H * W * Cin, nullptr, 0, dout, (size_t)N * 4 * H * W * Cout, &am));
    if (am) { g_am.in0 = am; g_am.w = am + 2; }
    return convt_wgrad(in, dout, dw, db, part, N, H, W, Cin, Cout, as_stream(stream));
}

extern "C" int eld_maxpool2x2_forward(const float* in, float* out, int N, int Ho, int Wo, int C, void* stream) {
    if (N == 0) return 0;
    if (!in || !out || C % 4) return ELD_EINVAL;
    return launch_maxpool_fwd(in, out, N, Ho, Wo, C, as_stream(stream));
}

extern "C" int eld_maxpool2x2_backward(const float* act, const float* dp, const float* skip, float* g, int N, int Ho, int Wo, int C, void* stream) {
    if (N == 0) return 0;
    if (!act || !dp || !g || C % 4) return ELD_EINVAL;
    return launch_maxpool_bwd(act, dp, skip, g, N, Ho, Wo, C, as_stream(stream));
}
