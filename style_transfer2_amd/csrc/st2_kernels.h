// Internal launch interface between the engine (engine.cpp) and the gfx950 kernels (*.hip).
// Everything here is device-pointer based; all launches go to the stream given.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace st2 {

constexpr int kMaxPartials = 1024;   // grid cap (= partial-sum slots) of every reducing kernel
constexpr int kConvCC = 4;           // K granularity of the packed conv weights (staged chunk = 4 or 8)
constexpr int kCoutQuantum = 64;     // conv weights are zero-padded to a multiple of this many outputs

// ------------------------------------------------------------------------------------------
// conv3x3 (pad 1, stride 1) as implicit GEMM on v_mfma_f32_32x32x2_f32.  NCHW fp32.
//   forward : out = relu(conv(in, W) + bias)                         (prototxt Convolution+ReLU)
//   dgrad   : out = mask(conv(in, W')) + inject   with W' = flipped/transposed W, where
//             mask(v) = (mask_src > 0 ? v : 0) when mask_src != nullptr  (ReLU backward of the
//             in-place blob below) and inject is the unmasked diff injected at that blob.
// Weights are pre-packed by pack_conv_weights(): [ceil(K/CC)][9][CC][MPad].
// ------------------------------------------------------------------------------------------
struct ConvProblem {
    const float* in;        // [K][H][W]   (K = input channels of this GEMM)
    const float* wpack;     // packed weights
    const float* bias;      // [MPad] or nullptr (forward only)
    float* out;             // [M][H][W]
    const float* mask_src;  // [M][H][W] or nullptr (dgrad only)
    const float* inject;    // [M][H][W] or nullptr (dgrad only)
    int K, M, MPad, H, W;
    int relu;               // forward epilogue
    unsigned long long* stamps = nullptr;   // diagnostic configs only
    unsigned short* out16 = nullptr;        // optional (direct kernel): bf16 channel-blocked copy of `out`, [M/8][H][W][8], M % 8 == 0
    float* pool_out = nullptr;              // optional (Winograd forward, see conv_wino_can_pool): also write maxpool2x2/2 of `out`
    unsigned char* pool_amap = nullptr;     // optional, with pool_out: [M][ph][pw] bytes = first-max slot | (max > 0) << 2 (launch_maxpool_bwd_amap)
    // optional (Winograd data-gradient directly below a max-pool, conv_wino_can_unpool): `in` is then the POOLED diff [K][H/2][W/2] and
    // unpool_amap the pool's arg-max map [K][H/2][W/2] (ConvProblem::pool_amap of the forward); the launch unpools while it stages
    const unsigned char* unpool_amap = nullptr;
    float* scratch = nullptr;               // optional: room for split-K partial sums (Winograd launches with few workgroups)
    size_t scratch_floats = 0;
};
size_t conv_pack_floats(int K, int M);                       // floats in a packed weight buffer
int conv_mpad(int M);
// host-side packers (plain CPU loops; weights arrive once per worker start)
void pack_conv_weights_fwd(const float* w /*M=Cout,K=Cin,3,3*/, int Cout, int Cin, float* dst);
void pack_conv_weights_dgrad(const float* w /*Cout,Cin,3,3*/, int Cout, int Cin, float* dst);
hipError_t launch_conv3x3(const ConvProblem& p, hipStream_t s);
// explicit tile configuration (measurement hook); cfg < 0 = heuristic
int conv_num_configs();
const char* conv_config_name(int cfg);
int conv_pick_config(const ConvProblem& p);
hipError_t launch_conv3x3_cfg(const ConvProblem& p, int cfg, hipStream_t s);
// Winograd F(2x2,3x3) variant of launch_conv3x3 (same ConvProblem, p.wpack = the Winograd pack, MPad unused)
size_t wino_pack_floats(int K, int M);
void pack_wino_weights_fwd(const float* w, int Cout, int Cin, float* dst);
void pack_wino_weights_dgrad(const float* w, int Cout, int Cin, float* dst);
bool conv_wino_ok(int K, int M, int H, int W);
// dx = pool backward of dy through the arg-max map a Winograd forward wrote (ConvProblem::pool_amap), ReLU mask of the pooled-from conv
// blob included (bit 2); H even, W % 4 == 0.  352 instead of 603 bytes moved per 64 outputs: neither the conv blob nor the pooled one is read.
hipError_t launch_maxpool_bwd_amap(const float* dy, const unsigned char* amap, float* dx, int C, int H, int W, hipStream_t s);
bool conv_wino_pool_amap_ok(int K, int M, int H, int W);   // ... and such a launch fills ConvProblem::pool_amap
bool conv_wino_can_unpool(int K, int M, int H, int W);   // a data-gradient launch of this shape may take ConvProblem::unpool_amap
bool conv_wino_can_skip_out(int K, int M, int H, int W);   // a forward launch with pool_out + pool_amap may pass out == nullptr
bool conv_wino_can_pool(int K, int M, int H, int W);   // launch_conv3x3_wino may fuse the following max-pool (ConvProblem::pool_out)
int conv_wino_splits(int K, int M, int H, int W);   // split-K factor the automatic path would use (1 = none); needs splits*M*H*W floats of scratch
hipError_t launch_conv3x3_wino(const ConvProblem& p, hipStream_t s);
hipError_t launch_conv3x3_wino_cfg(const ConvProblem& p, int variant, hipStream_t s);
hipError_t launch_wino_combine(const float* scratch, int splits, const float* bias, int relu, const float* mask_src, const float* inject,
                               float* out, int M, int H, int W, hipStream_t s);
// Split-operand Winograd (conv3x3_wino_split.hip; st_set_conv_algo(ctx, 2)): the same ConvProblem, p.wpack = the split pack (bf16 triples
// of G g G^T in A-fragment order) passed as const float*; fp32 results from six bf16 partial products per transform-domain product.
// K % 16 == 0, M % 64 == 0, W % 4 == 0; fused pool + arg-max map + skipped blob and split-K as launch_conv3x3_wino, no unpool_amap.
size_t wino_split_pack_elems(int K, int M);                  // bf16 elements
void pack_wino_split_weights_fwd(const float* w, int Cout, int Cin, unsigned short* dst);
void pack_wino_split_weights_dgrad(const float* w, int Cout, int Cin, unsigned short* dst);
bool conv_wino_split_ok(int K, int M, int H, int W);
int conv_wino_split_splits(int K, int M, int H, int W);
bool conv_wino_split_can_pool(int K, int M, int H, int W);
bool conv_wino_split_pool_amap_ok(int K, int M, int H, int W);
bool conv_wino_split_can_skip_out(int K, int M, int H, int W);
hipError_t launch_conv3x3_wino_split(const ConvProblem& p, hipStream_t s);   // 0: 128 ch x 4x32 px, 1: 64 ch x 8x32 px
// conv1_1-style dgrad (tiny M): direct VALU kernel, w is the ORIGINAL (Cout,Cin,3,3) layout
// conv1_1's data gradient on the matrix cores (conv3x3_dgrad_first.hip): Z = A @ dy (1x1, 9 M rows) + 9 M shifted adds; Cin <= 3
bool conv_dgrad_first_ok(int Cout, int Cin, int H, int W, bool bf16);
hipError_t launch_conv3x3_dgrad_first(const float* dy, const float* w, float* dx, const float* inject, int Cout, int Cin, int H, int W, hipStream_t s);
bool conv_dgrad_first_strip_ok(int Cout, int Cin, int H, int W);                      // fp32, Cout = 64: the strip walker (each row of the diff read once)
hipError_t launch_conv3x3_dgrad_first_strip(const float* dy, const float* w, float* dx, const float* inject, int Cout, int Cin, int H, int W, hipStream_t s);
bool conv_dgrad_first_quad_ok(int Cout, int Cin, int H, int W, const float* dy);      // fp32, W % 4 == 0: 16-byte operand loads
hipError_t launch_conv3x3_dgrad_first_quad(const float* dy, const float* w, float* dx, const float* inject, int Cout, int Cin, int H, int W, hipStream_t s);
hipError_t launch_conv3x3_dgrad_first16(const unsigned short* dy16, const float* w_rounded, float* dx, const float* inject, int Cout, int Cin,
                                        int H, int W, hipStream_t s);
// Forward of the first conv for the bf16 feature path (conv3x3_first_split.hip): fp32 operands split into three bf16 terms each, six exact
// partial products on the bf16 matrix cores, fp32 accumulation -- fp32-grade results, HBM-bound.  wpk = pack_conv_first_split's output (the bias rides in it as a 28th tap).  Cin = 3.
// Writes the fp32 blob (out, may be nullptr) and / or the bf16 channel-blocked copy (out16, may be nullptr).
size_t conv_first_split_pack_elems(int Cout);
void pack_conv_first_split(const float* w /*Cout, Cin, 3, 3*/, const float* bias /*Cout or nullptr*/, int Cout, int Cin, unsigned short* dst);
bool conv_first_split_ok(int Cin, int Cout, int H, int W);
hipError_t launch_conv3x3_first_split(const float* x, const unsigned short* wpk, float* out, unsigned short* out16,
                                      int Cin, int Cout, int H, int W, int relu, hipStream_t s, unsigned short* bits_out = nullptr);
bool conv_dgrad_smallM_ok(int Cout, int Cin);
hipError_t launch_conv3x3_dgrad_smallM(const float* dy, const float* w, float* dx, const float* inject,
                                       int Cout, int Cin, int H, int W, hipStream_t s);
// bf16 feature path: dy as bf16 channel-blocked copy [Cout/8][H][W][8] (Cout % 8 == 0), weights already rounded to bf16 values
hipError_t launch_conv3x3_dgrad_smallM16(const unsigned short* dy16, const float* w_rounded, float* dx, const float* inject,
                                         int Cout, int Cin, int H, int W, hipStream_t s);

// ------------------------------------------------------------------------------------------
// bf16 feature path (BASELINE config 3): conv operands in bf16 (v_mfma_f32_32x32x16_bf16), fp32 accumulate and
// fp32 blobs.  Activations are read from a channel-blocked bf16 copy [C/8][H][W][8]; out16 (optional) is the copy
// of this launch's output for the next conv.  Weight packs: [ceil(K/16)][9][2][MPad][8] bf16.
// ------------------------------------------------------------------------------------------
struct Conv16Problem {
    const unsigned short* in16; const unsigned short* wpack16; const float* bias;
    float* out;                             // fp32 [M][H][W]; nullptr: not written (the only consumer reads out16 / the pooled copy)
    unsigned short* out16; const float* mask_src; const float* inject;
    int K, M, MPad, H, W, relu;
    const unsigned short* mask16 = nullptr; // dgrad: ReLU mask from the bf16 copy of the blob below (instead of mask_src)
    // forward, optional (conv16_can_pool): max-pool 2x2/2 (Caffe ceil mode, first-max) of this launch's output
    unsigned short* pool16 = nullptr;       // bf16 channel-blocked pooled copy [M/8][ph][pw][8]
    float* pool32 = nullptr;                // fp32 pooled blob [M][ph][pw]
    unsigned char* amap = nullptr;          // [M/8][ph][pw][8] bytes: bits 0-1 arg-max slot (row-major in the window), bit 2 maximum > 0
    // data-gradient, optional: the style gradient of the blob this launch differentiates rides on the launch,
    //     out = mask(conv) + D' @ F        (D' = sw / norm * c2 * D as hi + lo bf16 terms: launch_style_fuse_pack; F = s_in16, M channels)
    const unsigned short* s_in16 = nullptr; const unsigned short* s_wpack16 = nullptr;
    // data-gradient directly below a max-pool, optional (conv16_can_unpool): in16 is the POOLED diff [K/8][H/2][W/2][8] and unpool_amap
    // the pool's arg-max map; the launch expands them in its staged tile (maxpool_bwd_idx16_k and its output are not needed)
    const unsigned char* unpool_amap = nullptr;
    // one bit per element of a post-ReLU blob, "its bf16 copy is non-zero", in the accumulator layout of the 32 x 32 MFMA tile:
    //     [M / 32][H * W][2] 16-bit words; word (blk, pixel, half), bit 8 h + e  <->  channel 32 blk + 4 half + 16 h + (e & 3) + 8 (e >> 2)
    // (conv16_bits_elems words, M % 32 == 0).  bits_out: written by a forward launch next to out16; mask_bits: read by the data
    // gradient instead of mask16 -- 1/16 of its bytes, one 2-byte load per 16 elements instead of two 8-byte ones
    unsigned short* bits_out = nullptr; const unsigned short* mask_bits = nullptr;
    unsigned long long* stamps = nullptr;   // tools/probes only: 6 words per workgroup: 100 MHz ticks at start / first chunk landed / main loop done / end, shader cycles at the two middle points
};
inline size_t conv16_bits_elems(int M, size_t hw) { return (size_t)(M / 32) * hw * 2; }
// A-operand image of the scaled D for the fused style term: [M / 16][hl][k half][MPad] quads
size_t style_fuse_pack_elems(int C, int MPad);
hipError_t launch_style_fuse_pack(const float* D, int ld, int C, int MPad, float c2, float sw, const float* norm, unsigned short* A16, hipStream_t s);
// partial sums of || c2 * D @ F ||^2 = c2^2 * n * sum_ij (D (D + A))_ij D_ij  (F F^T = n (D + A), n = C * hw): the trace value of the
// style gradient without materialising it; *n_partial partials in `partial`
int style_s2_trace_blocks(int C);
hipError_t launch_style_s2_trace(const float* D, int ld, const float* A, int C, double n, float c2, float* partial, int* n_partial, hipStream_t s);
bool conv16_can_pool(const Conv16Problem& p);
bool conv16_can_unpool(const Conv16Problem& p);
// dx16 = pool backward of dy16 through the arg-max map (all channel-blocked, C % 8 == 0), ReLU mask of the pooled-from blob included
hipError_t launch_maxpool_bwd_idx16(const unsigned short* dy16, const unsigned char* amap, unsigned short* dx16, int C, int H, int W, hipStream_t s);
size_t conv16_pack_elems(int K, int M);
void pack_conv_weights16_fwd(const float* w, int Cout, int Cin, unsigned short* dst);
void pack_conv_weights16_dgrad(const float* w, int Cout, int Cin, unsigned short* dst);
hipError_t launch_conv3x3_bf16(const Conv16Problem& p, hipStream_t s);
hipError_t launch_pack_act16(const float* src, unsigned short* dst, int C, size_t hw, hipStream_t s);

// ------------------------------------------------------------------------------------------
// max pool 2x2 stride 2, Caffe ceil mode, first-max arg-max (recomputed in backward).
// ------------------------------------------------------------------------------------------
int pooled_size(int n);
hipError_t launch_maxpool_fwd(const float* in, float* out, int C, int H, int W, hipStream_t s);
// dx = mask(scatter(dy at argmax)) + inject ; x is the pool INPUT blob (also the mask source)
hipError_t launch_maxpool_bwd(const float* dy, const float* x, float* dx, const float* inject,
                              int apply_mask, int C, int H, int W, hipStream_t s);

// ------------------------------------------------------------------------------------------
// Gram matrix  G = F F^T / n  (F is [C][hw]) as split-K MFMA GEMM + deterministic slab reduce.
// ------------------------------------------------------------------------------------------
struct GramPlan { int bt, tiles, splits, kslab; size_t slab_floats; };
GramPlan gram_plan(int C, int hw);
// bf16 feature path (gram16.hip): partials from the bf16 channel-blocked copy [C/8][hw][8] on the bf16 matrix cores, fp32
// accumulation; same slab format (gram_reduce finishes it).  Needs C % 8 == 0, hw % 64 == 0 and a gram_plan16 plan.
GramPlan gram_plan16(int C, int hw);
bool gram16_ok(int C, int hw, const GramPlan& pl);
// region of interest of a blob [C][H][W]: hw (= rw * rows) pixels starting at (y0, x0); pitch = W, plane = H*W
struct GramRoi { int y0, x0, rw, pitch; size_t plane; };
bool gram16_roi_ok(int C, int hw, size_t plane, const GramPlan& pl);
hipError_t launch_gram16_partial(const unsigned short* F16, float* slabs, int C, int hw, const GramPlan& pl, hipStream_t s,
                                 const GramRoi* roi = nullptr);
hipError_t launch_gram_partial(const float* F, float* slabs, int C, int hw, const GramPlan& pl, hipStream_t s,
                               const GramRoi* roi = nullptr);
// out[i][j] = sum_s slabs[s][i][j] / n  - (target ? target[i][j] : 0);  partial[blockIdx] = sum out^2
// `folded` is scratch of gram_fold_groups(pl) * C * C floats (two-stage reduction when there are many splits)
int gram_fold_groups(const GramPlan& pl);
// out = sum_s slabs[s] / divisor - (target ? target : 0)   (divisor = C*hw for a Gram matrix, 1 for raw sums)
hipError_t launch_gram_reduce(const float* slabs, float* folded, const float* target, float* out, int out_ld, float* partial,
                              int* n_partial, int C, double divisor, const GramPlan& pl, hipStream_t s);

// ------------------------------------------------------------------------------------------
// Style gradient  S = c2 * (D @ F)  (D = G - G_style, C x C symmetric; F = blob [C][H][W]) on the conv
// pipeline with a single tap.  Dp is D with leading dimension conv_mpad(C).
//   fused = 0 : dst = S                              partial[block] = sum S^2   (first evaluation)
//   fused = 1 : dst = (sw / *norm) * S + (accumulate ? dst : 0) ; partial as above
// ------------------------------------------------------------------------------------------
int style_grad_blocks(int C, int H, int W);          // partial-sum slots the launch writes
struct PixRoi { int y0, x0, y1, x1; };               // half-open pixel rectangle of a blob
hipError_t launch_style_grad(const float* Dp, const float* F, float* dst, float c2, int fused, float sw, const float* norm,
                             int accumulate, float* partial, int* n_partial, int C, int H, int W, hipStream_t s,
                             const PixRoi* roi = nullptr);
// bf16 feature path: the same S on the bf16 matrix cores, F read from its bf16 channel-blocked copy [C/8][hw][8], D split
// on the device into hi + lo bf16 terms (A16 = scratch of style_grad16_pack_elems(C) bf16).  C % 64 == 0.
bool style_grad16_ok(int C, size_t hw);
size_t style_grad16_pack_elems(int C);
int style_grad16_blocks(int C, size_t hw);
hipError_t launch_style_grad16(const float* Dp, int ld, unsigned short* A16, const unsigned short* F16, float* dst, float c2, int fused,
                               float sw, const float* norm, int accumulate, float* partial, int* n_partial, int C, size_t hw, hipStream_t s,
                               const GramRoi* roi = nullptr);   // roi: hw = its pixel count; reads / writes only those pixels of the blob
// out[0] = sum(part[0..n)) in double, rounded to float (deterministic, one workgroup)
hipError_t launch_sum_partials(const float* part, int n, float* out, hipStream_t s);
// inject = (sw / *norm) * S + (accumulate ? inject : 0)
hipError_t launch_scaled_accumulate(const float* S, float* inject, float sw, const float* norm,
                                    int accumulate, size_t n, hipStream_t s);

// ------------------------------------------------------------------------------------------
// per-layer elementwise terms (content / deep-dream), worker.py:249-256,271-277
// ------------------------------------------------------------------------------------------
struct LayerElemArgs {
    const float* feat;      // F
    const float* target;    // F_c (content features) or nullptr
    float* inject;          // written (overwritten) when write != 0
    size_t n;
    float cn_coef, dn_coef; // float(2/n), float(-2/n)
    float cw, dw;
    int content, deepdream, write;
    const float* norm_c;    // device scalars (valid when write != 0)
    const float* norm_d;
    float* part_d2;         // sum (F-Fc)^2
    float* part_gc2;        // sum (cn_coef (F-Fc))^2
    float* part_f2;         // sum F^2
    float* part_gd2;        // sum (dn_coef F)^2
    // region of interest (tile-sharded mode): sums and non-zero writes only inside; w == 0 means whole tensor
    int h, w, ry0, rx0, ry1, rx1;
};
hipError_t launch_layer_elem(const LayerElemArgs& a, int* n_partial, hipStream_t s);

// norm = sqrt(float(sum(partials) / n))
hipError_t launch_finalize_norm(const float* partial, int n_partial, double n, float* norm, hipStream_t s);

// ------------------------------------------------------------------------------------------
// image-space pass: TV + p-norm + combine (+ Adam), utils.py:285-304, worker.py:279-297,
// optimizers.py:20-27.  x/x_out are (3,H,W); periodic borders.
// ------------------------------------------------------------------------------------------
struct ImagePassArgs {
    const float* x;         // current image
    const float* scd;       // dL/dx from the network (may be nullptr => zeros)
    float* grad;            // combined gradient out (nullptr to skip)
    int C, H, W;
    float tv_w, tv_beta, p_w, p_pow;
    // Adam (enabled when x_out != nullptr)
    float* x_out; float* m; float* v;
    float d1, c1, d2, c2, corr1, corr2, step;
    int m_is_zero;          // m treated as 0 (after clear) without a memset
    int v_is_zero;
    float* partial;         // 6 rows of kMaxPartials: tv, p, scd^2, (tv_w g_tv)^2, (p_w g_p)^2, grad^2
    const float* dyn = nullptr;   // optional device {corr1, corr2, step} overriding the by-value ones (hipGraph replays)
};
hipError_t launch_set_scalars3(float* dst, float a, float b, float c, hipStream_t s);
hipError_t launch_image_pass(const ImagePassArgs& a, int* n_partial, hipStream_t s);
// Tile-sharded variant: the tile [ty, ty+th) x [tx, tx+tw) of a window image of pitch ww; neighbours outside
// the tile come from `ring` ([3][th+2][tw+2], the periodic-wrap neighbourhood gathered from the owners).
struct ImageTileArgs {
    ImagePassArgs base;      // x / scd / x_out / m / v are WINDOW tensors (3, wh, ww); C/H/W = 3, wh, ww
    const float* ring;
    int ty, tx, th, tw;
};
hipError_t launch_image_pass_tile(const ImageTileArgs& a, int* n_partial, hipStream_t s);

// Tile-sharded mode: the rectangles exchanged with ONE neighbour, packed into / unpacked from one contiguous buffer
constexpr int kMaxStripRects = 12;
struct StripTable { int n; int total; int y0[kMaxStripRects], x0[kMaxStripRects], h[kMaxStripRects], w[kMaxStripRects], off[kMaxStripRects]; };
// mode 0: buf = pack(tensor rects); 1: tensor rects = buf; 2: tensor rects += buf   (tensor is (C, wh, ww))
hipError_t launch_strip_copy(float* tensor, float* buf, const StripTable& t, int C, int wh, int ww, int mode, hipStream_t s);

// Pillow-exact separable resampling of float planes (utils.py:130-160).  Tables live on the device:
// lo[i] = first source index, n[i] = window length, k[i*kmax + j] = normalised double coefficients.
struct ResampleTable { const int* lo; const int* n; const double* k; int kmax; };
hipError_t launch_resample(const float* src, float* tmp, float* dst, int planes, int h_in, int w_in, int h_out, int w_out,
                           const ResampleTable& tx, const ResampleTable& ty, int clamp0, hipStream_t s);

// pre/deprocess, worker.py:63-71
hipError_t launch_preprocess_u8(const uint8_t* hwc, float* nchw, int H, int W, hipStream_t s);
hipError_t launch_preprocess_f32(const float* hwc, float* nchw, int H, int W, hipStream_t s);
hipError_t launch_deprocess(const float* nchw, float* hwc, int H, int W, hipStream_t s);

// ------------------------------------------------------------------------------------------
// trace finalisation: one tiny kernel turns the partial sums of an opfunc into the trace scalars
// ------------------------------------------------------------------------------------------
constexpr int kMaxTraceLayers = 24;
constexpr int kLayerSlots = 6;       // d2, gc2, f2, gd2, D2, S2
constexpr int kImageSlots = 6;
struct TraceLayer {
    int content, style, deepdream;
    float cw, sw, dw;
    double n;            // C*h*w
    double gram_n;       // C*C
    int count[kLayerSlots];
    const float* part[kLayerSlots];
    const float* norm;   // [3] c, s, d
};
struct TraceArgs {
    int n_layers;
    TraceLayer layer[kMaxTraceLayers];
    const float* image_part;     // kImageSlots rows of kMaxPartials
    int image_count;
    double image_n;              // 3*H*W
    float tv_w, p_w, p_pow;
    int have_grad;
    float* out;                  // [n_layers*6 + 8]
    double* sums;                // scratch: n_layers*6 + 6 doubles
};
hipError_t launch_finalize_trace(const TraceArgs& a, hipStream_t s);

// ------------------------------------------------------------------------------------------
// Fixed-step L-BFGS as a device-resident state machine (lbfgs.hip), optimizers.py:49-125, utils.py:29-46
// ------------------------------------------------------------------------------------------
constexpr int kLbfgsCorr = 10;                  // n_corr (optimizers.py:52)
constexpr int kLbfgsSlots = kLbfgsCorr + 1;     // ring slots: n_corr pairs + the one being formed
struct LbfgsDev {                               // lives in device memory; zero = empty history
    int count;                                  // pairs kept
    int free_slot;                              // slot of the pair being formed
    int order[kLbfgsSlots];                     // slot ids of the kept pairs, oldest first
    int pad_;
    double sy[kLbfgsSlots], yy[kLbfgsSlots];    // s.y and y.y per slot
    double alpha[kLbfgsSlots];                  // first-loop coefficients of the current recursion
    double last_sy;                             // s.y of the last candidate pair (kept or not)
};
// Gram form of the same recursion (lbfgs.hip, second half): the search direction lives in the span of the kept s and y
// vectors and the gradient, so the two-loop recursion runs on 2 k + 1 COEFFICIENTS against the matrix of their inner
// products; the vectors are streamed twice per step (one pass of inner products, one linear combination) instead of 4 k times.
constexpr int kLbNB = 2 * kLbfgsSlots + 1;      // basis ids: s of slot i = i, y of slot i = kLbfgsSlots + i, the gradient = 2 kLbfgsSlots
constexpr int kLbGramRows = 2 * kLbNB;          // partial-sum rows of the inner-product pass: g.b (kLbNB) then y.b (kLbNB)
struct LbfgsGram {                              // lives in device memory
    double B[kLbNB][kLbNB];                     // inner products of the live basis vectors (entries of dead ids are stale, never read)
    double delta[kLbNB];                        // inv_hv(g) = sum delta[id] b_id
    double r0;                                  // empty history: inv_hv(g) = g / r0
};
struct LbfgsVecs { float* s[kLbfgsSlots]; float* y[kLbfgsSlots]; };
struct LbfgsArgs {
    LbfgsVecs v;
    LbfgsDev* st;
    LbfgsGram* gm;          // Gram form only
    float* gpart;           // Gram form only: [kLbGramRows][kMaxPartials]
    float* part;            // [2][kMaxPartials] ping-pong partial sums of the chained dot products
    float* part2;           // [2][kMaxPartials] partial sums of s.y and y.y of the candidate pair
    const float* g;         // gradient at the current x (self.grad)
    float* p;               // work vector of the recursion
    float* x;               // the iterate, updated in place
    size_t n;
    float step;
    int apply;              // 1: the last link forms s = -step p and x += s; 0 (test hook): p = H g is left in `p`
    // tile-sharded mode (Gram form only): the vectors are this rank's TILE of the image, the inner products are summed over the ranks
    float* gsums;           // [kLbGramRows] this rank's sums of the pass (launch_lbfgs_gram_pass_local), all-reduced by the caller
    size_t n_global;        // elements of the WHOLE image (the unit-RMS first direction divides by it); 0: n
};
// p = inv_hv(g) by the two-loop recursion, every link one launch (axpy_i fused with dot_{i+1}); 24 launches, those
// beyond the current pair count return at once
hipError_t launch_lbfgs_two_loop(const LbfgsArgs& a, hipStream_t s);
// candidate pair in the free slot: mode 0: y = g_new - a.g; mode 1: y already stored.  Then the s.y > 1e-10 gate,
// the commit into the ring and the eviction of the oldest pair, all on the device.
hipError_t launch_lbfgs_pair(const LbfgsArgs& a, const float* g_new, int mode, hipStream_t s);
// Gram form.  launch_lbfgs_gram_pass: mode 0: inner products of a.g with the kept vectors (a fresh gradient, no candidate pair);
// mode 1: y = g_new - a.g into the free slot, inner products of y and of g_new with the kept vectors, the candidate s and y;
// then (one workgroup) the bookkeeping: rows of B, the s.y > 1e-10 gate / commit / eviction, and the coefficient recursion.
// launch_lbfgs_gram_apply: p = sum delta b, then s = -step p into the free slot and x += s (a.apply) or p left in a.p.
hipError_t launch_lbfgs_gram_pass(const LbfgsArgs& a, const float* g_new, int mode, hipStream_t s);
hipError_t launch_lbfgs_gram_apply(const LbfgsArgs& a, hipStream_t s);
// The same pass in two halves for a vector that is sharded over ranks: _local leaves this rank's kLbGramRows sums in a.gsums
// (fp32, as sdot returns them) instead of doing the bookkeeping; the caller all-reduces them; _global then runs the bookkeeping and
// the coefficient recursion on the summed values -- identically on every rank, so every rank derives the same direction.
hipError_t launch_lbfgs_gram_pass_local(const LbfgsArgs& a, const float* g_new, int mode, hipStream_t s);
hipError_t launch_lbfgs_gram_commit_global(const LbfgsArgs& a, int mode, hipStream_t s);
int lbfgs_gram_rows();      // kLbGramRows
// test hook support: B from a table of all pairwise inner products (float [kLbNB][kLbNB], row-major), then the recursion
hipError_t launch_lbfgs_gram_load(const LbfgsArgs& a, const float* dots, hipStream_t s);
// z = a*x + b*y  (host scalars; y may be nullptr)
hipError_t launch_lincomb(float a, const float* x, float b, const float* y, float* z, size_t n, hipStream_t s);
// tile-sharded L-BFGS: *out = sum a[i] b[i] over this rank's n elements (part: kMaxPartials floats of scratch); y = alpha x + y
hipError_t launch_vec_dot(const float* a, const float* b, size_t n, float* part, float* out, hipStream_t s);
hipError_t launch_vec_axpy(float alpha, const float* x, float* y, size_t n, hipStream_t s);
hipError_t launch_vec_div(double divisor, float* y, size_t n, hipStream_t s);      // y = float(double(y) / divisor)

}  // namespace st2
