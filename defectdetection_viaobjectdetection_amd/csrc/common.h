// Shared declarations for libmi355yolo.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace m355 {

typedef _Float16 half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float float4v __attribute__((ext_vector_type(4)));

constexpr int BK = 64;  // K elements per main-loop step of the implicit GEMM (one 128-byte LDS row)

// Arguments of the implicit-GEMM convolution kernel (all strides in ELEMENTS of the tensor's dtype).
struct ConvArgs {
  const half_t* x;     // input, NHWC fp16, already offset to the first input channel of the slice
  long x_bstride;      // elements between images
  int ldx;             // elements between pixels (total channels of the underlying buffer)
  int Hi, Wi, Cin;
  const half_t* w;     // packed weights [Cout_pad][Kpad], K = (kh*KS+kw)*Cin + cin
  int Kpad;
  const float* bias;   // [Cout_pad]
  void* y;             // output, fp16 (or fp32 if out_f32), offset to first output channel of the slice
  long y_bstride;
  int ldy;
  int Ho, Wo;          // output spatial size (for convT: equals Hi, Wi -- the GEMM's pixel grid)
  int Cout;            // logical output channels (for convT: 4*Co virtual channels)
  const half_t* res;   // optional residual (same pixel grid as y), offset to slice
  long r_bstride;
  int ldr;
  int ksize, stride, pad;
  int M;               // B*Ho*Wo
  int act;             // 1: SiLU
  int out_f32;         // 1: store fp32
  int convt_co;        // >0: ConvTranspose 2x2/s2 pixel-shuffle store with Co = convt_co
  const half_t* zero;  // >=16 bytes of zeros in device memory (source for padded taps)
  int tmode;           // 1: transposed-stride gather = dgrad of a 3x3 / stride-2 / pad-1 conv (x is dY, y is dX)
  int dbg;             // ablation switches for profiling experiments (0 in production)
  unsigned long long* stamps;  // diagnostic builds: per-block s_memtime stamps (nullptr in production)
  // Upsample read-through (1x1 convs only): input channels [0, csplit) are NOT in x but in the half-resolution tensor
  // x2 -- output pixel (h, w) reads x2 pixel (h >> 1, w >> 1), i.e. Upsample(2x, nearest) + Concat without the copy.
  const half_t* x2;
  long x2_bstride;
  int ldx2;
  int csplit;          // 0: off
  // Phase convolution (ksize 2, stride 1): the composition ConvTranspose(2x2, s2) -> Conv(3x3, s1, p1) is, for each of
  // the four output phases (py, px), a 2x2 convolution over the LOW-resolution input with window rows h-1+py .. h+py
  // (cols likewise).  Virtual channel q * convt_co + co, q = py * 2 + px, pixel-shuffle store like ConvTranspose; the
  // bias is a [9][convt_co] table indexed by the output pixel's border class (first / interior / last row x column).
  int phase;
  // Phase conv + trailing 1x1 conv (proto.cv3) in the epilogue: y2 = act(w2 . act(phase conv) + bias2), cout2 = 32 output
  // channels, K = convt_co = 128 (one 128 x 128 tile holds every channel of a pixel).  w2: fp16 [32][128] row-major in
  // LOGICAL channel order; the output goes to y (ldy / y_bstride describe the 32-channel tensor).
  const half_t* w2;
  const float* bias2;
  int cout2;
  // Head output conv + decode in one launch (fp32 1x1 conv whose single 128-channel tile holds the whole 64 + nc + nm row
  // of a pixel): the DFL expectation, dist2bbox, the class sigmoid and the coefficient copy of head_decode_kernel run in
  // the epilogue and write the prediction rows; the raw map is written too only when dec_keep_raw is set.
  float* dec_preds;        // (B, dec_A, 4 + nc + nm) fp32; nullptr: off
  int dec_A, dec_level_off, dec_nc, dec_nm, dec_keep_raw;
  float dec_stride;
  // Rows of the packed weight buffer `w` and floats of `bias` the caller allocated (0: conv_cout_pad(Cout), the
  // engine's padding).  Every launcher refuses a channel tile whose ceil(Cout / tile) * tile rows exceed it: the
  // kernels fetch whole weight-row tiles by LDS-DMA with no per-row bound.
  int w_rows;
  // Tile queue of the persistent kernels: two ints in device memory, zero before the first launch (the kernel re-arms
  // them): [0] tiles claimed beyond each block's static first one, [1] blocks that have left.  nullptr: static walk.
  // One queue per op -- never shared by launches that can run at the same time.
  int* tileq;
  // Fragment-ordered copy of `w` for the weights-in-registers kernels (conv1x1_wreg, conv3x3_s2c64, ...): fragment
  // f at halves [f * 512, f * 512 + 512), lane-linear 16 bytes = the A operand of one 32x32x16 MFMA.  nullptr: the kernels
  // gather the fragments from `w` (64 scattered 16-byte loads per fragment).
  const half_t* wf;
  const half_t* wf2;   // the same for `w2` (proto_phase_wreg.hip: the eight fragments of proto.cv3)
  int bias_lds;        // set by the implicit-GEMM launcher: this many bias floats are staged in LDS behind the stages (0: none)
};

// tile ids for launch_conv_igemm(force_tile)
enum { TILE_AUTO = -1, TILE_128x128 = 0, TILE_64x128 = 1, TILE_32x256 = 2, TILE_64x256 = 3, TILE_64x128W8 = 5, TILE_HALO = 16, TILE_HALO8W = 17, TILE_HALO4W = 18, TILE_HALOWIDE = 19, TILE_C32 = 20, TILE_SLAB = 25, TILE_M32 = 26, TILE_M32_128 = 27, TILE_M32_64x16 = 28, TILE_M32_64x8 = 29, TILE_W1 = 32, TILE_PLANES = 33 };

// Experiment switches (environment variables M355_*), read once per process: launchers are on the hot path.
struct Knobs {
  bool no_fast_epi, no_wide, no_persist, stem_gather, no_m32, static_tiles, no_bias_lds;
  int persist, halo_variant, smallm;
};
const Knobs& knobs();

int launch_conv_igemm(const ConvArgs& a, int force_tile, hipStream_t s);
// tile heuristic: cout = (virtual) output channels, M = output pixels of the whole batch
int conv_pick_tile(int cout, long M);
// rows the packed weight buffer must be padded to for a given Cout (multiple of the channel tile)
int conv_cout_pad(int cout);
int conv_kpad(int cin, int ksize);
// true when a launch with channel tile `bch` stays inside the weight / bias rows the caller provided
inline bool conv_rows_covered(const ConvArgs& a, int bch) {
  const int need = (a.Cout + bch - 1) / bch * bch;
  return need <= (a.w_rows > 0 ? a.w_rows : conv_cout_pad(a.Cout));
}
// (channel, pixel) extent of a forced tile id of m355_conv2d_fwd; false for ids no launcher implements
bool conv_forced_tile_extent(int tile, int cout, int* bch, int* bpx);

// 3x3 stride-1 halo-tile kernel (conv3x3_halo.hip)
bool conv3x3_halo_ok(const ConvArgs& a);
int launch_conv3x3_halo(const ConvArgs& a, int variant, hipStream_t s);  // variant 0 auto, 1: 16x16 px / 8 waves, 2: 8x16 px / 4 waves, 3: wide
// 128 ch x 16x16 px, K depth 32 per step (conv3x3_wide.hip)
bool conv3x3_wide_ok(const ConvArgs& a);
int launch_conv3x3_wide(const ConvArgs& a, hipStream_t s);
// v_mfma_f32_32x32x16_f16 halo kernel (conv3x3_m32.hip); which: 0 = by shape, 1 = <128 ch, 8 rows>, 2 = <64, 16>, 3 = <64, 8>
bool conv3x3_m32_ok(const ConvArgs& a);
int launch_conv3x3_m32(const ConvArgs& a, int which, hipStream_t s);
// Conv3x3 / s2 (32 -> 64) + Conv1x1 (64 -> 64) in one persistent patch kernel (conv3x3_s2c32.hip): a.w2 / a.bias2 / a.cout2 set
bool conv_s2c32_cv1_ok(const ConvArgs& a);
int launch_conv_s2c32_cv1(const ConvArgs& a, hipStream_t s);
// narrow maps (W <= 26): slabs of full-width rows, linear pixel groups (conv3x3_small.hip)
bool conv3x3_slab_ok(const ConvArgs& a);
int launch_conv3x3_slab(const ConvArgs& a, hipStream_t s);
// D-FINE decoder ops (dfine_kernels.hip)
int launch_msda(const float* value, const float* loc, const float* attn, float* out, int B, int S, int H, int D, int Q, int P,
                int L, const int* shapes_hw, const int* points_per_level, int discrete, hipStream_t s,
                const float* ref = nullptr, float offset_scale = 0.f);
int launch_dfine_decode(const float* dist, const float* project, const float* ref, float* boxes, long n, int nbins1,
                        float reg_scale, int clamp01, hipStream_t s);
// 1x1, K <= 512, Cout % 128 == 0: weights in registers, persistent (conv1x1_wreg.hip)
bool conv1x1_wreg_ok(const ConvArgs& a);
int launch_conv1x1_wreg(const ConvArgs& a, hipStream_t s);
// The output convs of one head level + the decode of their rows in one launch (head_tail.hip): x = the level's 224-channel
// branch tensor (64 box | 128 class | 32 coefficient channels), dense rows; wf = 18 MFMA fragments (box 2 x 4, class 8,
// coefficients 2) of the block-diagonal weight matrix; bias = [64 | nc | nm]; preds = (B, A, 4 + nc + nm) fp32.
struct HeadTailArgs {
  const half_t* x;
  int ldx;             // 224
  long M;              // B * H * W pixels of the level
  int HW, W;           // anchors of the level per image, grid width
  float stride;
  int A, level_off, nc, nm;
  const half_t* wf;
  const float* bias;
  float* preds;
};
bool head_tail_ok(const HeadTailArgs& a);
int launch_head_tail(const HeadTailArgs& a, hipStream_t s);
// 3x3 / s2 (64 -> 128) + 1x1 (128 -> 128): model.3 + model.4.cv1 of the s scale, 3x3 weights in registers (conv3x3_s2c64.hip)
bool conv_s2c64_cv1_ok(const ConvArgs& a);
int launch_conv_s2c64_cv1(const ConvArgs& a, hipStream_t s);
// the composed Proto launch (phase conv + proto.cv3) with the weights in registers, one phase per block (proto_phase_wreg.hip)
bool proto_phase_wreg_ok(const ConvArgs& a);
int launch_proto_phase_wreg(const ConvArgs& a, hipStream_t s);
// Cin = Cout = 32, weights-stationary persistent halo kernel (conv3x3_c32.hip)
bool conv3x3_c32_ok(const ConvArgs& a);
int launch_conv3x3_c32(const ConvArgs& a, hipStream_t s);

// A whole C2f block body with 32 hidden channels in one launch (c2f_c32.hip): t = Conv3x3(y1), y2 = y1 + Conv3x3(t),
// out = Conv1x1([y0, y1, y2]); x holds [y0, y1] (what C2f.cv1 wrote) in its first 64 channels.
struct C2fC32Args {
  const half_t* x; long x_bstride; int ldx;      // NHWC fp16, channels [0, 32) = y0, [32, 64) = y1
  int H, W, B;
  const half_t *waf, *wbf;                       // fragment-ordered wa (plain rows) / wb (operand rows), 18 x 512 halves each; nullptr: gather
  const half_t *wa, *wb, *wc;                    // packed fp16 rows [32][kpad_a], [32][kpad_b] (K = tap * 32 + cin), [64][kpad_c] (K = y0, y1, y2)
  int kpad_a, kpad_b, kpad_c;
  const float *ba, *bb, *bc;                     // folded biases: 32, 32, 64 floats
  half_t* y; long y_bstride; int ldy;            // 64 output channels
  int shortcut;                                  // 1: y2 = y1 + ...
};
bool c2f_c32_ok(const C2fC32Args& a);
int launch_c2f_c32(const C2fC32Args& a, hipStream_t s);

// Row-slab 3x3 kernels (conv3x3_planes.hip): a whole Bottleneck -- y = [x +] act(conv3x3(act(conv3x3(x, wa) + ba), wb) + bb), hidden
// tensor in LDS -- or one 3x3 conv y = act(conv3x3(x, wb) + bb) [+ res].  Weights as MFMA A fragments in K-loop order
// (planes_frag_pack in engine.hip): fragment ((cb * NP + p) * 9 + tap) * 2 + s = rows 32 cb .. + 31 (plain row permutation),
// K = tap * Cin + 32 p + 16 s .. + 15; 1 KiB each, lane-linear.
struct PlanesArgs {
  const half_t* x; long x_bstride; int ldx;      // NHWC fp16 input slice (Cin channels)
  int H, W, B, Cin, Cout;                        // H, W: INPUT map
  int stride;                                    // 1, or 2 (single mode only: output map H / 2 x W / 2)
  const half_t *wfa, *wfb; int cblocks_a, cblocks_b;   // fragment-ordered weights and their 32-channel blocks; wfa / ba: first conv of a pair (nullptr in single mode)
  const float *ba, *bb;
  half_t* y; long y_bstride; int ldy;
  const half_t* res; long r_bstride; int ldr;    // optional residual (the shortcut of a Bottleneck: x itself)
  int act;
  unsigned long long* stamps;                    // diagnostic: 8 uint64 per wave (nullptr in production)
};
bool bneck_pair_shape_ok(int C, int H, int W);
bool bneck_pair_ok(const PlanesArgs& a);
int launch_bneck_pair(const PlanesArgs& a, hipStream_t s);
bool conv3x3_planes_ok(const PlanesArgs& a);
int launch_conv3x3_planes(const PlanesArgs& a, hipStream_t s);

struct StemArgs {
  const uint8_t* x; int B, H, W;     // uint8 NHWC (B,H,W,3)
  const half_t* w16;                 // [Cout][32] fp16: k = (kh*3+kw)*3+ci, rows 27..31 zero; NOT scaled by 1/255
  const float* bias;                 // [Cout]
  half_t* y; long y_bstride; int ldy; int Cout;
};
int launch_stem(const StemArgs& a, hipStream_t s);
// stem + model.1 + model.2.cv1 in one launch (conv_stem_s2c32.hip): a = the model.1 + cv1 launch, st = the stem launch
bool stem_s2c32_ok(const ConvArgs& a, const StemArgs& st);
int launch_stem_s2c32(const ConvArgs& a, const StemArgs& st, hipStream_t s);
// the same launch with team X building the next tile's patch image while team Y runs the two convolutions (conv_stem_c2.hip)
bool stem_s2c32_v2_ok(const ConvArgs& a, const StemArgs& st);
int launch_stem_s2c32_v2(const ConvArgs& a, const StemArgs& st, hipStream_t s);

int launch_sppf_pool(const half_t* x, long x_bstride, int ldx, half_t* y, long y_bstride, int ldy,
                     int B, int H, int W, int C, hipStream_t s);
int launch_sppf_pool_bwd(const half_t* a, long a_bs, int lda, const half_t* y, long y_bs, int ldy, const half_t* gy, long gy_bs,
                         int ldgy, half_t* ga, long ga_bs, int ldga, int B, int H, int W, int C, int accumulate, hipStream_t s);
int launch_upsample2x(const half_t* x, long x_bstride, int ldx, half_t* y, long y_bstride, int ldy,
                      int B, int H, int W, int C, hipStream_t s);
// ADown's pooling (yolov9c): a = avg_pool2d(x, 2, 1, 0)[..., :C/2] (H-1 x W-1), m = max_pool2d(avg_pool2d(x, 2, 1, 0)[..., C/2:], 3, 2, 1) (H/2 x W/2)
int launch_adown_pool(const half_t* x, long x_bstride, int ldx, half_t* a, long a_bstride, int lda, half_t* m, long m_bstride,
                      int ldm, int B, int H, int W, int C, hipStream_t s);
int launch_head_decode(const float* raw, int B, int in_h, int in_w, int nc, int nm, float* preds,
                       hipStream_t s);
int launch_nms(const float* preds, int B, int A, int nc, int nm, float conf, float iou, int max_det,
               float* dets, int* counts, void* workspace, size_t workspace_bytes, hipStream_t s);
size_t nms_workspace_bytes(int B, int A);
int launch_proto_masks(const float* dets, const int* counts, const half_t* protos, int B, int max_det,
                       int nm, int mh, int mw, int in_h, int in_w, uint8_t* masks, hipStream_t s);

// weight gradient (conv_wgrad.hip): dw fp32 [Cout][k*k*Cin] (KRSC).  Split-K partial slabs go to the caller's workspace
// (conv_wgrad_workspace_bytes) and are added in a fixed order: bitwise reproducible.  -3: workspace missing / too small.
size_t conv_wgrad_workspace_bytes(int B, int Ho, int Wo, int Cin, int Cout, int ksize);
// 3x3 / stride-1 layers from spatial patches (conv_wgrad3.hip); launch_conv_wgrad routes to it and adds the slabs
bool conv_wgrad3_ok(int B, int H, int W, int Cin, int Cout, int ksize, int stride, int pad, int lddz, int ldx);
size_t conv_wgrad3_workspace_bytes(int B, int H, int W, int Cin, int Cout);
int launch_conv_wgrad3(const half_t* dz, long dz_bs, int lddz, const half_t* x, long x_bs, int ldx, int B, int H, int W, int Cin,
                       int Cout, float* dw, const half_t* zero, float* ws, size_t ws_bytes, int* splitk, hipStream_t s);
int launch_conv_wgrad(const half_t* dz, long dz_bstride, int lddz, const half_t* x, long x_bstride, int ldx, int B,
                      int Hi, int Wi, int Cin, int Ho, int Wo, int Cout, int ksize, int stride, int pad, float* dw,
                      const half_t* zero, float* ws, size_t ws_bytes, hipStream_t s);
// floats of workspace the train-mode batch-norm launches need for C channels (per-block partial sums + ticket)
size_t bn_workspace_floats(int C);
// gradient glue of the training step (train_kernels.hip): fixed-order column sums, nearest-2x upsample backward, input conversion
long colsum_workspace_floats(long nb, int cols);
int launch_colsum(const void* src, int src_f16, long nb, long bstride, long rows, int ld, int cols, float* ws, float* out, hipStream_t s);
// stride-2 input gradient with 32 forward input / 64 output channels as a wave-private chunk stream (conv_dgrad_s2c32.hip; a = the phase-2 form
// of the im2col kernel: x = dY, y = dX, w = the phase-form packed matrix)
bool dgrad_s2c32_ok(const ConvArgs& a);
int launch_dgrad_s2c32(const ConvArgs& a, hipStream_t s);
// YOLOv9c training glue (train_kernels.hip): RepConvN's SiLU(a + b) and ADown's pooling front, forward and backward
int launch_addsilu_fwd(const half_t* a, const half_t* b, half_t* v, half_t* y, long npix, int ldy, int C, hipStream_t s);
int launch_addsilu_bwd(const half_t* v, const half_t* dy, int lddy, half_t* g, long npix, int C, hipStream_t s);
int launch_adown_fwd(const half_t* x, long x_bs, int ldx, half_t* p1, long p1_bs, int ld1, half_t* p2, long p2_bs, int ld2, unsigned char* arg,
                     int B, int H, int W, int c, hipStream_t s);
int launch_adown_bwd(const half_t* g1, long g1_bs, int ld1, const half_t* g2, long g2_bs, int ld2, const unsigned char* arg, half_t* gx,
                     long gx_bs, int ldg, int B, int H, int W, int c, int accumulate, hipStream_t s);
int launch_upsample2x_bwd(const half_t* g, long g_bs, int ldg, half_t* d, long d_bs, int ldd, int B, int H, int W, int C, int accumulate,
                          hipStream_t s);
int launch_u8_to_f16x8(const unsigned char* src, half_t* dst, long npx, hipStream_t s);
// mask term of the segmentation loss + its gradients in one pass (loss_kernels.hip)
int launch_mask_loss(const float* coef, const void* protos, int protos_f16, const int* masks, const int* inst, const float* boxes,
                     const float* w, int B, int K, int mh, int mw, float* slot_sum, float* d_coef, void* d_protos, int d_protos_f16,
                     const float* gscale, hipStream_t s);
int launch_box_loss(const float* logits, const float* anchors, const float* targets, const float* weights, long n, float* box_term,
                    float* dfl_term, float* d_box, float* d_dfl, hipStream_t s);
int launch_dfl_decode(const float* raw, long rows, int A, int rw, int nc, const float* anchors, const float* strides, float* boxes,
                      float* scores, hipStream_t s);
int launch_tal_assign(const float* scores, const float* boxes, const float* anchors_px, const int* gt_cls, const float* gt_boxes,
                      const unsigned char* gt_valid, int B, int A, int G, int nc, void* ws, float* t_boxes, float* t_scores,
                      unsigned char* fg, long* gt_idx, hipStream_t s);
int launch_repack(const void* d_jobs /* m355_repack_job[] (include/mi355yolo.h) */, const int* d_block_job, int nblocks, hipStream_t s);

// train-mode BatchNorm + SiLU (train_kernels.hip)
int launch_bn_silu_train_fwd(const half_t* z, long npix, int ldz, int C, const float* gamma, const float* beta,
                             float eps, half_t* y, int ldy, const half_t* res, int ldr, float* sums, float* mean_out,
                             float* invstd_out, int act, float* run_mean, float* run_var, float momentum, hipStream_t s);
int launch_adamw_step(float* p, const float* g, float* m, float* v, float* ema, const unsigned char* group, long n,
                      float lr, float lr_bias, float beta1, float beta2, float eps, float wd, int step, float grad_mul,
                      float ema_d, hipStream_t s);
int launch_sgd_step(float* p, const float* g, float* buf, float* ema, const unsigned char* group, long n, float lr,
                    float lr_bias, float momentum, int nesterov, float wd, float grad_mul, float ema_d, hipStream_t s);
int launch_grad_sumsq(const float* g, long n, float* out, hipStream_t s);   // out: grad_sumsq_workspace_floats() floats
size_t grad_sumsq_workspace_floats();
// mosaic + affine + HSV + flip gather (augment.hip); params: device array of B m355_aug_params
int launch_augment(const uint8_t* cache, const void* params, uint8_t* out, int B, int H, int W, hipStream_t s);
int launch_bn_silu_train_bwd(const half_t* z, const half_t* dy, long npix, int ldz, int lddy, int C, const float* mean,
                             const float* invstd, const float* gamma, const float* beta, float* rsum, half_t* dz,
                             int lddz, int act, float* ws, hipStream_t s);


// ---------------------------------------------------------------------------------------------------------
// Fast conv epilogue shared by the conv kernels (device code, include from .hip files only).
// A wave measured 13-15 k cycles in the generic epilogue of a 64 ch x 128 px tile: ~85 instructions per 16-byte
// store (per-group validity branches with EXEC save / restore, 64-bit address multiplies, an LDS / global bias read
// with its own wait) at one instruction per 4 cycles per wave.  When the whole wave tile is inside the tensor the
// addresses are affine (base + nt * ystep + group * 32 channels), the bias sits in 16 registers and the SiLU is
// packed: ~40 instructions per store, no branch.
// ---------------------------------------------------------------------------------------------------------
#ifdef __HIPCC__
// No FMA contraction in the epilogue arithmetic: the generic and the fast epilogue must give the same bits, so that an
// image's result does not depend on whether its tile was a full one (batch size / position invariance is tested).
__device__ __forceinline__ float m355_silu(float v) {
#pragma clang fp contract(off)
  const float e = __builtin_amdgcn_exp2f(v * -1.4426950408889634f);
  return v * __builtin_amdgcn_rcpf(1.0f + e);
}

// Round-to-fp16 of an epilogue value.  The value is made opaque first: otherwise the compiler may fuse the last
// multiply (or the residual add) with the conversion into v_fma_mix*_f16 for SOME elements of SOME template
// instantiations -- one rounding instead of two, a rare 1-ulp difference that breaks bit-exact tile / batch invariance.
__device__ __forceinline__ half_t m355_to_half(float v) {
  asm volatile("" : "+v"(v));
  return (half_t)v;
}

// acc[2s][nt] / acc[2s+1][nt] hold channels 8g..8g+3 / 8g+4..8g+7 of group s for pixel nt (see conv_igemm.hip);
// yp / rp point at (pixel nt = 0, group 0) of this lane; ystep / rstep = elements between consecutive nt.
template <int MT, int NT, bool ACT, bool RES>
__device__ __forceinline__ void conv_epilogue_fast(float4v (&acc)[MT][NT], const float4v (&bias)[MT / 2][2], half_t* yp,
                                                   long ystep, const half_t* rp, long rstep) {
#pragma clang fp contract(off)
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
    for (int s = 0; s < MT / 2; ++s) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        v[j] = acc[2 * s][nt][j] + bias[s][0][j];
        v[4 + j] = acc[2 * s + 1][nt][j] + bias[s][1][j];
      }
      if (ACT) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = m355_silu(v[j]);
      }
      if (RES) {
        const half8 rv = *(const half8*)(rp + nt * rstep + s * 32);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += (float)rv[j];
      }
      half8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = m355_to_half(v[j]);
      *(half8*)(yp + nt * ystep + s * 32) = o;
    }
  }
}
// Same, with one output pointer per pixel tile (ConvTranspose pixel-shuffle stores: not affine in nt), no residual.
template <int MT, int NT, bool ACT>
__device__ __forceinline__ void conv_epilogue_fast_ptrs(float4v (&acc)[MT][NT], const float4v (&bias)[MT / 2][2],
                                                        half_t* const (&yp)[NT]) {
#pragma clang fp contract(off)
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
    for (int s = 0; s < MT / 2; ++s) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        v[j] = acc[2 * s][nt][j] + bias[s][0][j];
        v[4 + j] = acc[2 * s + 1][nt][j] + bias[s][1][j];
      }
      if (ACT) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = m355_silu(v[j]);
      }
      half8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = m355_to_half(v[j]);
      *(half8*)(yp[nt] + s * 32) = o;
    }
  }
}
#endif


}  // namespace m355
