/*
 * mi355yolo.h -- C-ABI of libmi355yolo.so: the MI355X (gfx950) native YOLOv8-seg hot path.
 *
 * The reference (CSMaus/DefectDetection_viaObjectDetection) has no FFI of its own: its hot path is the
 * Python call `YOLO(w).predict(src, save=True)` / `.train(...)` into the un-vendored `ultralytics`
 * package (BscanBased/yolo8_seg_predict.py:5-9, BscanBased/yolo_seg_train.py:7-19).  This header is the
 * boundary a maintainer binds instead (ctypes stub in INTEGRATION.md); each entry point names the
 * upstream stage it replaces (SURVEY.md section 8a row ids A3..A12).
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; no C++ / torch types; no exceptions cross the boundary.
 *   - return 0 on success, negative m355_status on error; text via m355_last_error().
 *   - every device buffer is CALLER-OWNED (e.g. a PyTorch-ROCm tensor's data_ptr()); the library owns
 *     only the weights + workspace inside the engine handle.
 *   - every call is ASYNCHRONOUS on the caller's stream (`stream` is a hipStream_t passed as void*),
 *     no hidden synchronisation -- except the entry points marked [sync], which exist for unit parity.
 *   - one engine per device; an engine is not thread-safe; distinct engines are.
 *   - there is NO CPU fallback: without a gfx950 device m355_create fails with M355_ERR_NO_DEVICE.
 */
#ifndef MI355YOLO_H_
#define MI355YOLO_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct m355_engine m355_engine;

typedef enum {
  M355_OK = 0,
  M355_ERR_INVALID = -1,    /* bad argument / shape */
  M355_ERR_NO_DEVICE = -2,  /* no gfx950 device visible */
  M355_ERR_HIP = -3,        /* HIP runtime error (text in last_error) */
  M355_ERR_STATE = -4,      /* weights not loaded, batch too large, ... */
  M355_ERR_NOMEM = -5
} m355_status;

/* Model description: replaces `YOLO("yolov8{n,s,m,l,x}-seg.yaml")` graph construction
 * (yolo_seg_train.py:7).  nc = number of classes (data-seg.yaml:4-5 -> 1). */
typedef struct {
  int scale;      /* 'n','s','m','l','x' */
  int nc;         /* classes */
  int in_h, in_w; /* network input size, multiples of 32 (640x640 headline) */
  int max_batch;  /* workspace is sized for this many images */
} m355_model_desc;

/* One convolution of the graph, in canonical order (the order weights are supplied in). */
typedef struct {
  char name[64];   /* ultralytics state-dict prefix, e.g. "model.2.m.0.cv1" (conv+bn) or
                      "model.22.cv2.0.2" (plain conv2d with bias) or "model.22.proto.upsample" */
  int cin, cout;   /* logical channels */
  int k, stride;   /* kernel size (1,2,3), stride */
  int has_bn;      /* 1: Conv2d(bias=False)+BN+SiLU (fold BN before m355_set_conv_weights) */
  int transposed;  /* 1: ConvTranspose2d(k=2,s=2,bias) -- weight layout (cin,cout,2,2) */
  int act;         /* 1: SiLU epilogue */
} m355_conv_info;

/* Version / build info string (static storage). */
const char* m355_version(void);

/* Last error text for this engine (or the last global error when e == NULL). */
const char* m355_last_error(const m355_engine* e);

/* Build the layer plan + allocate weights/workspace on the current HIP device.           [sync] */
int m355_create(const m355_model_desc* desc, m355_engine** out);
void m355_destroy(m355_engine* e);

/* Graph introspection (lets the host map a state dict onto the engine). */
int m355_num_convs(const m355_engine* e);
int m355_get_conv_info(const m355_engine* e, int idx, m355_conv_info* out);
int m355_num_anchors(const m355_engine* e);           /* 8400 at 640x640 */
int m355_pred_width(const m355_engine* e);            /* 4 + nc + 32 */
int m355_proto_hw(const m355_engine* e, int* h, int* w); /* 160x160 at 640 */
size_t m355_workspace_bytes(const m355_engine* e);
double m355_flops_per_image(const m355_engine* e);    /* 2 * conv MACs */

/* Supply BN-folded fp32 weights of conv `idx` (host pointers, copied; caller keeps ownership):
 * w is (cout,cin,k,k) [or (cin,cout,2,2) when transposed], bias is (cout).  Replaces upstream
 * `fuse()` + `.to(device)`.                                                  [sync] */
int m355_set_conv_weights(m355_engine* e, int idx, const float* w, const float* bias);

/* Inference forward: d_in is uint8 NHWC (B,in_h,in_w,3) letterboxed pixels in
 * [0,255] (the /255 normalisation is folded into the stem conv).  Outputs:
 *   d_preds  float32 (B, A, 4+nc+32): [cx,cy,w,h (pixels), class scores (sigmoid), 32 mask coefs]
 *   d_protos float16 (B, H/4, W/4, 32) NHWC
 * This is upstream's (B,4+nc+32,A) / (B,32,H/4,W/4) pair in anchor-major / NHWC order. */
int m355_forward(m355_engine* e, const void* d_in_u8_nhwc, int batch, float* d_preds, void* d_protos,
                 void* stream);

/* ---- measurement hooks (bench.py's live roofline figure) ----
 * One "op" = one kernel launch of the forward plan.  With profiling enabled m355_forward brackets every
 * launch with hipEventRecord on the caller's stream; m355_collect_op_times synchronises those events
 * and returns, per op, the accumulated milliseconds and number of launches since profiling was enabled. */
typedef struct {
  char kernel[48];        /* kernel family, e.g. "conv_igemm<128x128,k3>" */
  char layer[64];         /* first state-dict prefix the launch computes */
  double flops_per_image; /* algorithmic FLOPs (2*MACs), 0 for non-conv ops */
  double bytes_per_image; /* algorithmic activation bytes: inputs read once + outputs written once */
  double weight_bytes;    /* weight bytes, read once per launch */
} m355_op_info;
int m355_num_ops(const m355_engine* e);
int m355_get_op_info(const m355_engine* e, int idx, m355_op_info* out);
int m355_set_profiling(m355_engine* e, int enable);
int m355_collect_op_times(m355_engine* e, double* ms_sum, long* counts);                  /* [sync] */

/* Train-mode style raw head maps: float32 (B, A, 64+nc+32) = [box DFL logits(64),
 * class logits(nc), mask coefs(32)] before decode.  Valid after m355_forward on the same stream. */
int m355_get_raw_head(m355_engine* e, const float** d_raw, int* width);
/* Asynchronous device-to-device copy of the first `batch` images of the raw head maps into d_out. */
int m355_copy_raw_head(m355_engine* e, int batch, float* d_out, void* stream);
/* keep != 0 (the default of a new engine): the raw maps are written and the decode is its own launch.  keep == 0 (the
 * predict path): each head level's output convs and the decode of its rows run as one launch (csrc/head_tail.hip; same
 * arithmetic, bit-identical predictions), the raw maps are NOT written and m355_get_raw_head / m355_copy_raw_head fail with
 * M355_ERR_STATE.  (Models whose head does not fit that kernel -- nm != 32, nc > 32 -- always write the raw maps.) */
int m355_set_keep_raw(m355_engine* e, int keep);

/* Post-processing: batched NMS + mask assembly.
 *   d_dets   float32 (B, max_det, 6+32)  rows [x1,y1,x2,y2,conf,cls,coefs] in letterboxed pixels,
 *            sorted by confidence descending
 *   d_counts int32 (B)
 *   d_masks  uint8 (B, max_det, in_h, in_w) binary masks (may be NULL to skip mask assembly) */
int m355_postprocess(m355_engine* e, const float* d_preds, const void* d_protos, int batch, float conf,
                     float iou, int max_det, float* d_dets, int* d_counts, uint8_t* d_masks, void* stream);

/* ---- Environment switches ----------------------------------------------------------------------------------------------------
 * The library is configured by m355_model_desc and the arguments of each call.  The M355_* environment variables below are
 * read-once A/B switches for measurements and tests; none is needed in production and none changes the ABI.  Unset = the default
 * described here.  Switches that change which kernels m355_forward launches (results stay within the parity tolerances; every
 * pair is A/B-tested in tests/test_c2f_fused_gpu.py / test_engine_gpu.py):
 *   M355_NO_PAIR        Bottleneck pairs as two launches instead of one bneck_pair launch (conv3x3_planes.hip)
 *   M355_PAIR64=1       also use the pair launch for 64-channel bottlenecks (measured no gain: off)
 *   M355_NO_PLANES      the 20 x 20 level on conv3x3_slab instead of the row-slab kernel
 *   M355_NO_C2F32       model.2 as three launches instead of c2f_c32
 *   M355_NO_W1_SPLIT    model.15.cv1 (Upsample + Concat read through) on the im2col kernel instead of conv1x1_wreg's split form
 *   M355_NO_PLANES_S2, M355_NO_PLANES_M64   the stride-2 3x3 convs / the 64 -> 64 conv of the 40 x 40 head level off the row-slab kernel
 *   M355_NO_DGRAD_PHASES  (training) stride-2 input gradients as the masked nine-tap gather instead of four phase convs
 *   M355_NO_WGRAD_STEM    (training) layer 0's weight gradient on the pixel-axis GEMM instead of wgrad_stem_kernel
 *   M355_NO_WGRAD_S2C32, M355_NO_DGRAD_S2C32   (training) model.1's weight / input gradient on the pixel-axis GEMM / the im2col kernel
 *   M355_NO_W1, M355_NO_S2C64, M355_NO_S2C32, M355_NO_PROTOR, M355_NO_PROTOFUSE(3), M355_NO_HEADTAIL, M355_NO_STEMFUSE,
 *   M355_NO_STEM2, M355_NO_CVFUSE, M355_NO_UPFUSE, M355_NO_C32, M355_NO_M32, M355_NO_WIDE, M355_NO_HALO, M355_NO_SLAB,
 *   M355_DECFUSE=1      each falls back from one fused / specialised kernel to the kernels it replaced (names = file names in csrc/)
 *   M355_NO_LANES, M355_LANE_PLAN, M355_SUBBATCH(_OPS), M355_NO_SUBBATCH   stream lanes / sub-batches inside a forward
 * Tuning knobs of single kernels (tile queues, slot counts, priorities): M355_PERSIST, M355_NO_PERSIST, M355_STATIC_TILES,
 *   M355_M32_SLOTS, M355_C32_SLOTS, M355_C32_WASTE, M355_WIDE_SLOTS, M355_WIDE_STAGGER, M355_HALO_VARIANT, M355_K1_TILE, M355_SMALLM,
 *   M355_S2C32_RING, M355_S2C64_PRIO, M355_PROTOR_PRIO, M355_C2F_NOPRIO, M355_STEM2_NXB, M355_NO_FAST_EPI, M355_NO_BIAS_LDS,
 *   M355_STEM_GATHER, M355_MASK_TILE, M355_WGRAD_BLOCKS, M355_WGRAD3_{BLOCKS,MINTILES,SHRINK,SLABMB}, M355_NO_WGRAD3,
 *   M355_NO_TRAIN_W1, M355_NO_TRAIN_C32.
 * Diagnostics (write files / print timings, never set in production): M355_STAMPS, M355_*_STAMPS, M355_*_DBG, M355_BNECK_REPS;
 * testing only: M355_HEADTAIL_MAXM (forces head levels off the head_tail launch).
 * Python side (train_engine.py, bench.py): M355_NO_WGRAD_STREAM, M355_NO_HEAD_STREAM, M355_SIDE_STREAMS, M355_NO_LOSS_KERNELS,
 * M355_DIST_BACKEND, M355_DIST_SAME_DEVICE. */

/* ---- per-op entry points for unit parity (all device pointers, asynchronous unless noted) ---- */

/* NHWC fp16 convolution + bias (+SiLU) (+residual) via the implicit-GEMM MFMA kernel.
 * h_w fp32 (cout,cin,k,k) and h_bias fp32 (cout) are HOST pointers, packed + uploaded here.  [sync] */
int m355_conv2d_fwd(const void* d_x_f16_nhwc, int B, int H, int W, int cin, const float* h_w,
                    const float* h_bias, int cout, int k, int stride, int act,
                    const void* d_res_f16_nhwc, void* d_y_f16_nhwc, int out_f32, int force_tile,
                    void* stream);
/* A whole C2f block body with 32 hidden channels in ONE launch (csrc/c2f_c32.hip; upstream nn.modules.block.C2f with
 * n = 1 after its cv1):  t = SiLU(conv3x3(y1; wa) + ba);  y2 = SiLU(conv3x3(t; wb) + bb) (+ y1 if shortcut);
 * out = SiLU(conv1x1([y0, y1, y2]; wc) + bc).  d_x fp16 NHWC (B,H,W,64) = [y0, y1]; d_y fp16 NHWC (B,H,W,64);
 * h_wa / h_wb fp32 (32,32,3,3), h_wc fp32 (64,96,1,1), biases fp32: HOST pointers (BN folded), packed + uploaded
 * here.  H % 8 == 0 and W % 16 == 0.  [sync] */
int m355_c2f_c32_fwd(const void* d_x_f16_nhwc, int B, int H, int W, const float* h_wa, const float* h_ba,
                     const float* h_wb, const float* h_bb, const float* h_wc, const float* h_bc, int shortcut,
                     void* d_y_f16_nhwc, void* stream);
/* A whole C2f Bottleneck (upstream nn.modules.block.Bottleneck with e = 1.0: cv1 3x3 -> cv2 3x3, optional shortcut)
 * in ONE launch (csrc/conv3x3_planes.hip): t = SiLU(conv3x3(x; wa) + ba) stays in LDS (rounded to fp16 exactly where the
 * two-launch form stores it), y = SiLU(conv3x3(t; wb) + bb) (+ x if shortcut).  d_x fp16 NHWC (B,H,W,ldx) of which the first C
 * channels are x; d_y fp16 NHWC (B,H,W,ldy), first C channels written; C in {64, 128}; h_wa / h_wb fp32 (C,C,3,3) and
 * biases fp32 (C): HOST pointers (BN folded), packed + uploaded here.  Shapes whose row slabs do not fit LDS are refused
 * (M355_ERR_INVALID).  [sync] */
int m355_bneck_pair_fwd(const void* d_x_f16_nhwc, int B, int H, int W, int C, int ldx, const float* h_wa, const float* h_ba,
                        const float* h_wb, const float* h_bb, int shortcut, void* d_y_f16_nhwc, int ldy, void* stream);
/* Per-op parity entries of the fused launches (round 3 kernels; host weights fp32 with BN folded, packed + uploaded here exactly as
 * m355_set_conv_weights packs them; a shape the kernel does not take is refused with M355_ERR_INVALID).  [sync]
 *   m355_s2c64_cv1_fwd       csrc/conv3x3_s2c64.hip: Conv3x3/s2 (64 -> 128) + SiLU -> fp16 -> Conv1x1 (128 -> 128) + SiLU; upstream
 *                            model.3 + model.4.cv1 of yolov8s.  d_x (B,H,W,64) -> d_y (B,H/2,W/2,128); H/2 and W/2 multiples of 8.
 *   m355_stem_s2c32_cv1_fwd  csrc/conv_stem_c2.hip (two_team = 1) / conv_stem_s2c32.hip (0): uint8 image -> Conv3x3/s2 (3 -> 32, the
 *                            1/255 input scale inside) -> Conv3x3/s2 (32 -> 64) -> Conv1x1 (64 -> 64), SiLU + fp16 after each; upstream
 *                            model.0 + model.1 + model.2.cv1.  d_in (B,H,W,3) uint8 -> d_y (B,H/4,W/4,64); H/4 % 8 == 0, W/4 % 16 == 0.
 *   m355_proto_phase_fwd     csrc/proto_phase_wreg.hip: Proto.upsample (ConvTranspose2d 2x2/s2, bias; weight (cin,cout,2,2)) -> Proto.cv2
 *                            (3x3 + SiLU) -> Proto.cv3 (1x1, 128 -> 32, + SiLU) composed into four 2x2 phase convs.  d_x (B,H,W,128) ->
 *                            d_y (B,2H,2W,32); H % 8 == 0, W % 16 == 0.
 *   m355_head_tail_fwd       csrc/head_tail.hip: the three output convs of one Detect / Segment level (cv2.l.2 64 -> 64 box bins,
 *                            cv3.l.2 128 -> nc, cv4.l.2 32 -> 32, with bias) + DFL + dist2bbox + sigmoid: d_x (B,H,W,224) = [box 64 |
 *                            class 128 | coefficient 32] branch tensor -> rows [level_off, level_off + H W) of d_preds (B,A,4+nc+32) fp32. */
int m355_s2c64_cv1_fwd(const void* d_x_f16_nhwc, int B, int H, int W, const float* h_w3, const float* h_b3, const float* h_w1,
                       const float* h_b1, void* d_y_f16_nhwc, void* stream);
int m355_stem_s2c32_cv1_fwd(const void* d_in_u8_nhwc, int B, int H, int W, const float* h_w0, const float* h_b0, const float* h_w1,
                            const float* h_b1, const float* h_w2, const float* h_b2, void* d_y_f16_nhwc, int two_team, void* stream);
int m355_proto_phase_fwd(const void* d_x_f16_nhwc, int B, int H, int W, const float* h_wt, const float* h_bt, const float* h_w3,
                         const float* h_b3, const float* h_wc, const float* h_bc, void* d_y_f16_nhwc, void* stream);
int m355_head_tail_fwd(const void* d_x_f16_nhwc, int B, int H, int W, int nc, float stride, const float* h_w2, const float* h_b2,
                       const float* h_w3, const float* h_b3, const float* h_w4, const float* h_b4, float* d_preds, int A, int level_off,
                       void* stream);
/* Data gradient of Conv2d(k in {1,3}, stride in {1,2}, pad k/2, no bias) (SURVEY A13 backward): dY fp16 NHWC
 * (B,Ho,Wo,cout) -> dX fp16 NHWC (B,H,W,cin).  Runs on the same implicit-GEMM kernel: stride 1 = convolution
 * with the spatially flipped, channel-transposed weights; stride 2 = four 2x2 phase convs over dY on even maps (tmode 2 of
 * m355_conv_launch), the masked transposed-stride gather otherwise (or with M355_NO_DGRAD_PHASES=1).  h_w is the
 * FORWARD weight fp32 (cout,cin,k,k) on the host.                                                   [sync] */
int m355_conv2d_dgrad(const void* d_dy_f16_nhwc, int B, int H, int W, int cin, const float* h_w, int cout, int k,
                      int stride, void* d_dx_f16_nhwc, void* stream);
/* Weight gradient of Conv2d(k in {1,3}, stride in {1,2}, pad k/2) (SURVEY A13 backward): X fp16 NHWC (B,H,W,cin),
 * dY fp16 NHWC (B,Ho,Wo,cout) -> dW fp32 DEVICE buffer in KRSC order (cout, k, k, cin) = the packed forward
 * weight order.  Deterministic: split-K partial slabs in a workspace the call allocates, added in a fixed order. [sync] */
int m355_conv2d_wgrad(const void* d_x_f16_nhwc, const void* d_dy_f16_nhwc, int B, int H, int W, int cin, int cout,
                      int k, int stride, float* d_dw_krsc, void* stream);
/* Train-mode BatchNorm2d (batch statistics, biased variance, eps) + optional SiLU on fp16 NHWC (B,H,W,C)
 * (SURVEY A13): y = act(gamma * (z - mean) * invstd + beta).  gamma/beta/mean/invstd/ws are DEVICE fp32 arrays;
 * d_ws is a workspace of m355_bn_workspace_floats(C) floats (per-block partial sums; a second small kernel adds them in
 * block order: no float atomics, bitwise reproducible); d_mean/d_invstd receive the saved statistics. */
size_t m355_bn_workspace_floats(int C);
int m355_bn_silu_train_fwd(const void* d_z, int B, int H, int W, int C, const float* d_gamma, const float* d_beta,
                           float eps, int act, void* d_y, float* d_mean, float* d_invstd, float* d_ws, void* stream);
/* Backward of the above: dz (fp16 NHWC) and d_dbeta_dgamma (device float[2*C]: [0:C] = dbeta, [C:2C] = dgamma). */
int m355_bn_silu_train_bwd(const void* d_z, const void* d_dy, int B, int H, int W, int C, const float* d_mean,
                           const float* d_invstd, const float* d_gamma, const float* d_beta, int act, void* d_dz,
                           float* d_dbeta_dgamma, float* d_ws, void* stream);
/* ConvTranspose2d(k=2,s=2)+bias; h_w fp32 (cin,cout,2,2).                                   [sync] */
int m355_convt2x2_fwd(const void* d_x_f16_nhwc, int B, int H, int W, int cin, const float* h_w,
                      const float* h_bias, int cout, void* d_y_f16_nhwc, void* stream);
/* Stem conv: uint8 NHWC (B,H,W,3) -> fp16 NHWC (B,H/2,W/2,cout), 3x3 s2 p1 + bias + SiLU;
 * h_w fp32 (cout,3,3,3) is applied to pixel/255.                                            [sync] */
int m355_stem_fwd(const void* d_in_u8, int B, int H, int W, const float* h_w, const float* h_bias, int cout,
                  void* d_y_f16_nhwc, void* stream);
/* SPPF pooling: x fp16 NHWC (B,H,W,C) -> y (B,H,W,3C) = [mp5(x), mp5(mp5(x)), mp5^3(x)]. */
int m355_sppf_pool(const void* d_x, int B, int H, int W, int C, void* d_y, void* stream);
/* Nearest 2x upsample, fp16 NHWC (B,H,W,C) -> (B,2H,2W,C). */
int m355_upsample2x(const void* d_x, int B, int H, int W, int C, void* d_y, void* stream);
/* Head decode: raw (B,A,64+nc+32) f32 -> preds (B,A,4+nc+32) f32 for an in_h x in_w input. */
int m355_head_decode(const float* d_raw, int B, int in_h, int in_w, int nc, float* d_preds, void* stream);
/* Batched NMS only. preds (B,A,4+nc+nm). */
int m355_nms(const float* d_preds, int B, int A, int nc, int nm, float conf, float iou, int max_det,
             float* d_dets, int* d_counts, void* stream);
/* Mask assembly only: dets (B,max_det,6+32), counts (B), protos fp16 (B,mh,mw,32)
 * -> masks uint8 (B,max_det,in_h,in_w). */
int m355_proto_masks(const float* d_dets, const int* d_counts, const void* d_protos, int B, int max_det,
                     int mh, int mw, int in_h, int in_w, uint8_t* d_masks, void* stream);

/* ---- raw strided launches (asynchronous; every pointer is a DEVICE pointer) --------------------------------
 * Used by the training orchestrator (defectdetection_viaobjectdetection_amd/train_engine.py) to run the kernels
 * on channel slices of shared NHWC buffers.  Strides are in ELEMENTS of the tensor's dtype. */
typedef struct {
  const void* x; int64_t x_bstride; int32_t ldx; int32_t hi, wi, cin;   /* input slice (fp16 NHWC) */
  const void* w_packed; int32_t kpad;   /* fp16 [ceil128(cout)][kpad], K = (kh*k+kw)*cin + ci, kpad % 64 == 0 */
  const float* bias;                    /* fp32 [ceil128(cout)] */
  void* y; int64_t y_bstride; int32_t ldy; int32_t ho, wo, cout;          /* output slice (fp16, or fp32 if out_f32) */
  const void* res; int64_t r_bstride; int32_t ldr;                        /* optional residual added after act */
  int32_t ksize, stride, pad, batch;
  int32_t act, out_f32, convt_co, tmode;
  const void* zero_page;                /* >= 16 zero bytes */
} m355_conv_args;
/* Convolution / dgrad / ConvTranspose forward on the implicit-GEMM or halo kernel (chosen by shape).
 * tmode 1: input gradient of a 3x3 / stride-2 / pad-1 conv by a transposed-stride gather (x = dY, y = dX, all nine taps masked per
 *          output parity; any size).
 * tmode 2: the same gradient as FOUR 2x2 phase convs over dY (one per parity class of the dX pixel): ksize = 2, stride = 1, pad = 0,
 *          hi x wi = ho x wo = the dY map, cout = 4 * convt_co virtual channels (phase-major), convt_co = the forward input channels,
 *          y = the (2 ho) x (2 wo) dX slice, w_packed = [4 * convt_co][(ty, tx, forward cout)] with zero rows for the taps a phase does
 *          not have (16 tap slots for 9 taps; tmode 1 multiplies 36).  Needs convt_co % 64 == 0, or 128 % convt_co == 0 and no res.
 *          With convt_co % 64 == 0 the K axis of a phase is COMPACT -- tap (ty, tx) of phase (a, b) at slot ty * (1 + b) + tx, zeros
 *          behind -- and the phase's K loop ends after its (1 + a)(1 + b) taps: 9 tap slots in all.  Otherwise slot = ty * 2 + tx. */
int m355_conv_launch(const m355_conv_args* a, void* stream);

typedef struct {
  const void* dz; int64_t dz_bstride; int32_t lddz;   /* dY slice (fp16 NHWC), spatial ho x wo, cout channels */
  const void* x; int64_t x_bstride; int32_t ldx;      /* forward input slice, spatial hi x wi, cin channels */
  int32_t hi, wi, cin, ho, wo, cout;
  int32_t ksize, stride, pad, batch;
  float* dw;                                          /* fp32 [cout][ksize*ksize*cin] (KRSC), overwritten */
  const void* zero_page;
  float* ws; int64_t ws_bytes;                        /* >= m355_wgrad_workspace_bytes(...): split-K partial slabs */
} m355_wgrad_args;
/* Deterministic (no float atomics): every K split stores its partial tile to its slab of `ws`, a second kernel adds the
 * slabs in a fixed order.  M355_ERR_INVALID when ws is missing / too small for the shape. */
size_t m355_wgrad_workspace_bytes(int32_t batch, int32_t ho, int32_t wo, int32_t cin, int32_t cout, int32_t ksize);
int m355_wgrad_launch(const m355_wgrad_args* a, void* stream);

/* Train-mode BN(+SiLU)(+residual) on slices: y = act(bn(z)) + res.  running_mean / running_var (may be NULL) get the
 * momentum update r = (1 - momentum) * r + momentum * batch_stat (unbiased variance), like torch BatchNorm2d. */
int m355_bn_train_fwd_launch(const void* z, int64_t npix, int32_t ldz, int32_t C, const float* gamma, const float* beta,
                             float eps, int32_t act, void* y, int32_t ldy, const void* res, int32_t ldr, float* mean,
                             float* invstd, float* ws, float* running_mean, float* running_var, float momentum,
                             void* stream);
int m355_bn_train_bwd_launch(const void* z, const void* dy, int64_t npix, int32_t ldz, int32_t lddy, int32_t C,
                             const float* mean, const float* invstd, const float* gamma, const float* beta, int32_t act,
                             void* dz, int32_t lddz, float* dbeta_dgamma, float* ws, void* stream);
int m355_sppf_pool_launch(const void* x, int64_t x_bstride, int32_t ldx, void* y, int64_t y_bstride, int32_t ldy,
                          int32_t B, int32_t H, int32_t W, int32_t C, void* stream);
/* Re-pack of fp32 master weights into fp16 GEMM layouts, every strided convert-copy of a training step in one launch.
 * A job copies a 4-d iteration space n[0..3] (n[3] fastest) from fp32 `src` to fp16 `dst` with signed ELEMENT strides on both
 * sides; block0 = index of the job's first block; block b of the launch handles elements [(b - block0) * 1024, +1024) of job
 * block_job[b].  Jobs and block_job live in DEVICE memory (built once: the pointers of a training engine never move). */
typedef struct m355_repack_job {
  const void* src;
  void* dst;
  int32_t n[4];
  int64_t ss[4];
  int64_t ds[4];
  int32_t block0;
  int32_t pad_;
} m355_repack_job;
int m355_repack_launch(const m355_repack_job* d_jobs, const int32_t* d_block_job, int32_t nblocks, void* stream);
/* Backward of the SPPF pooling chain (autograd of three F.max_pool2d(5, 1, 2) upstream): a = the pooled input slice, y = the
 * forward concat slice [y1 | y2 | y3] (3C channels), gy = its gradient, ga = d(loss)/da stored (accumulate = 0) or added to
 * what ga holds (fp16).  Gather formulation, no atomics: bitwise reproducible.  H * W * 96 bytes of LDS (<= 160 KB). */
int m355_sppf_pool_bwd_launch(const void* a, int64_t a_bstride, int32_t lda, const void* y, int64_t y_bstride, int32_t ldy,
                              const void* gy, int64_t gy_bstride, int32_t ldgy, void* ga, int64_t ga_bstride, int32_t ldga,
                              int32_t B, int32_t H, int32_t W, int32_t C, int32_t accumulate, void* stream);
/* Task-aligned assignment of the training loss (csrc/loss_kernels.hip; upstream TaskAlignedAssigner: top-10 by score^0.5 * CIoU^6 among the
 * anchors whose centre lies strictly inside the box, multiply-claimed anchors to the truth of highest CIoU, normalised alignment as
 * target score).  All pointers device: scores (B,A,nc) f32, boxes (B,A,4) xyxy px, anchors_px (A,2), gt_cls (B,G) int32, gt_boxes (B,G,4),
 * gt_valid (B,G) uint8; ws = 12 * B * G * 10 bytes of scratch; t_boxes (B,A,4), t_scores (B,A,nc), fg (B,A) uint8, gt_idx (B,A) int64 must
 * be ZERO on entry (only the positives are written).  Asynchronous on `stream`. */
int m355_tal_assign_launch(const float* scores, const float* boxes, const float* anchors_px, const int32_t* gt_cls, const float* gt_boxes,
                           const uint8_t* gt_valid, int32_t B, int32_t A, int32_t G, int32_t nc, void* ws, float* t_boxes, float* t_scores,
                           uint8_t* fg, int64_t* gt_idx, void* stream);
/* YOLOv9c training glue (csrc/train_kernels.hip; the graph /root/reference/BscanBased/yolo_seg_train.py:7 names), fp16 NHWC, asynchronous:
 *  addsilu: RepConvN's tail  y = SiLU(a + b)  of its two activation-free Conv + BN branches (upstream RepConvN.forward).  a, b, v, g are
 *    dense (npix, C) rows; v = fp16(a + b) is kept for the backward; y / dy are slices with row strides ldy / lddy.  Backward:
 *    g = dy * SiLU'(v), the gradient of BOTH branches (one buffer).
 *  adown: ADown's pooling front (upstream ADown.forward): t = avg_pool2d(x, 2, 1, 0); p1 = t[:, :c] (B,H-1,W-1,c);
 *    p2 = max_pool2d(t[:, c:], 3, 2, 1) (B,Ho,Wo,c), Ho = (H-2)/2+1; argmax (B,Ho,Wo,c) uint8 = window position 3*ky+kx of the first
 *    maximum in row-major order (torch's rule), written by the forward and read by the backward, which GATHERS: gx (B,H,W,2c) from the
 *    gradients g1 of p1 and g2 of p2 in a fixed order (no atomics); accumulate = 1 adds fp16(result) to what gx holds. */
int m355_addsilu_fwd_launch(const void* a, const void* b, void* v, void* y, int64_t npix, int32_t ldy, int32_t C, void* stream);
int m355_addsilu_bwd_launch(const void* v, const void* dy, int32_t lddy, void* g, int64_t npix, int32_t C, void* stream);
int m355_adown_fwd_launch(const void* x, int64_t x_bstride, int32_t ldx, void* p1, int64_t p1_bstride, int32_t ld1, void* p2,
                          int64_t p2_bstride, int32_t ld2, uint8_t* argmax, int32_t B, int32_t H, int32_t W, int32_t c, void* stream);
int m355_adown_bwd_launch(const void* g1, int64_t g1_bstride, int32_t ld1, const void* g2, int64_t g2_bstride, int32_t ld2,
                          const uint8_t* argmax, void* gx, int64_t gx_bstride, int32_t ldg, int32_t B, int32_t H, int32_t W, int32_t c,
                          int32_t accumulate, void* stream);
int m355_upsample2x_launch(const void* x, int64_t x_bstride, int32_t ldx, void* y, int64_t y_bstride, int32_t ldy,
                           int32_t B, int32_t H, int32_t W, int32_t C, void* stream);
/* Gradient glue of the training step (each replaces a strided torch expression reached from yolo_seg_train.py:12's backward):
 *  colsum: out[c] = sum over b < nb, r < rows of src[b * bstride + r * ld + c] for c < cols, fp32 result, fixed summation order
 *    (bias gradients: `d_raw[:, lo:hi, :].sum((0, 1))`, `gy.float().sum((0, 1, 2))`).  src fp32 (cols <= 256) or fp16 (src_f16 = 1:
 *    cols and ld multiples of 8, cols <= 2048); strides in elements; ws: m355_colsum_workspace_floats(nb, cols) floats.
 *  upsample2x_bwd: d (B,H,W,C) = or += the 2x2 sums of g (B,2H,2W,C) (fp32 sum of the four taps, one rounding to fp16;
 *    accumulate = 1 adds that fp16 value to d).  C, ldg, ldd multiples of 8.
 *  u8_to_f16x8: (npx, 3) uint8 -> (npx, 8) fp16 rows [r/255, g/255, b/255, 0 x5] = `(u8.float() / 255).half()`, the 8-channel
 *    input buffer of the training stem. */
int64_t m355_colsum_workspace_floats(int64_t nb, int32_t cols);
int m355_colsum_launch(const void* src, int32_t src_f16, int64_t nb, int64_t bstride, int64_t rows, int32_t ld, int32_t cols, float* ws,
                       float* out, void* stream);
int m355_upsample2x_bwd_launch(const void* g, int64_t g_bstride, int32_t ldg, void* d, int64_t d_bstride, int32_t ldd, int32_t B,
                               int32_t H, int32_t W, int32_t C, int32_t accumulate, void* stream);
int m355_u8_to_f16x8_launch(const uint8_t* src, void* dst, int64_t npx, void* stream);

/* Mask term of the segmentation loss with its gradients, forward and backward in one pass (replaces
 * v8SegmentationLoss.single_mask_loss and its autograd, reached from /root/reference/BscanBased/yolo_seg_train.py:12).
 * For slot k of image b (a foreground anchor): pred[p] = coef[b,k,:] . protos[b,p,:] over 32 channels,
 * slot_sum[b,k] = sum over the pixels p with x1 <= col < x2, y1 <= row < y2 of BCEWithLogits(pred[p], masks[b,p] == inst[b,k]);
 * with L = (1 / (mh mw)) sum_{b,k} weights[b,k] slot_sum[b,k]:  d_coef = dL/dcoef (B,K,32) fp32;  d_protos = g * dL/dprotos
 * (B,mh,mw,32), fp32 or fp16 (d_protos_f16 = 1), every element written, g = *gscale (a DEVICE scalar: the gradient arriving at L
 * in the caller's backward pass) or 1 when gscale is NULL.  slot_sum + d_coef and d_protos are two independent kernels: pass
 * slot_sum = d_coef = NULL or d_protos = NULL to run only one of them (forward: value and d_coef; backward: d_protos).
 * protos (B,mh,mw,32) NHWC fp16 (protos_f16 = 1) or fp32; masks (B,mh,mw) int32 overlap-encoded; boxes (B,K,4) x1,y1,x2,y2 in
 * prototype pixels; a slot with weight 0 is skipped (its outputs are 0).  No float atomics: bitwise reproducible. */
int m355_mask_loss_launch(const float* coef, const void* protos, int32_t protos_f16, const int32_t* masks, const int32_t* inst,
                          const float* boxes, const float* weights, int32_t B, int32_t K, int32_t mh, int32_t mw, float* slot_sum,
                          float* d_coef, void* d_protos, int32_t d_protos_f16, const float* gscale, void* stream);

/* Box (CIoU) and DFL terms of the loss on n foreground slots with their gradients w.r.t. the slot's 4 x 16 distribution logits,
 * one pass (replaces BboxLoss.forward / bbox_iou(CIoU=True) / DFLoss and their autograd, yolo_seg_train.py:12).  logits (n,4,16);
 * anchors (n,2) cell centres and targets (n,4) xyxy, both in grid units; weights (n), 0 = skip the slot (outputs 0).
 * box_term[s] = w (1 - CIoU(anchor -/+ E[softmax], target)), dfl_term[s] = w mean over the 4 sides of the two-bin cross entropy
 * at clamp(target distance, 0, 14.99); d_box / d_dfl (n,64) = d box_term / d logits, d dfl_term / d logits. */
int m355_box_loss_launch(const float* logits, const float* anchors, const float* targets, const float* weights, int64_t n,
                         float* box_term, float* dfl_term, float* d_box, float* d_dfl, void* stream);
/* Boxes and class scores of all anchors for the target assignment (no gradient): raw (rows, rw) head rows [64 DFL logits | nc class
 * logits | ...], row r belongs to anchor r % A -> boxes (rows,4) xyxy pixels = (anchor -/+ E[softmax over 16 bins]) * stride,
 * scores (rows,nc) = sigmoid.  anchors (A,2) grid units, strides (A). */
int m355_dfl_decode_launch(const float* raw, int64_t rows, int32_t A, int32_t rw, int32_t nc, const float* anchors, const float* strides,
                           float* boxes, float* scores, void* stream);

/* Optimizer step over a flat fp32 parameter buffer (replaces torch.optim.AdamW / SGD + ModelEMA.update reached from
 * /root/reference/BscanBased/yolo_seg_train.py:12).  group[i]: 0 decayed weights, 1 norm weights, 2 biases (lr_bias).
 * grad_mul = clip_coef / loss_scale.  ema may be NULL.  step counts from 1 (Adam bias correction). */
int m355_adamw_step(float* p, const float* g, float* m, float* v, float* ema, const uint8_t* group, int64_t n, float lr,
                    float lr_bias, float beta1, float beta2, float eps, float weight_decay, int32_t step, float grad_mul,
                    float ema_decay, void* stream);
int m355_sgd_step(float* p, const float* g, float* momentum_buf, float* ema, const uint8_t* group, int64_t n, float lr,
                  float lr_bias, float momentum, int32_t nesterov, float weight_decay, float grad_mul, float ema_decay,
                  void* stream);
/* out[0] = sum of squares of the finite entries of g, out[1] = number of non-finite entries.  `out` is a DEVICE buffer of
 * m355_grad_sumsq_workspace_floats() floats (the block partials sit behind the two results; a one-block kernel adds them
 * in block order): fixed reduction order, bitwise reproducible. */
size_t m355_grad_sumsq_workspace_floats(void);
int m355_grad_sumsq(const float* g, int64_t n, float* out, void* stream);

/* Training-time augmentation on the device image cache (replaces upstream's CPU Mosaic / RandomPerspective /
 * RandomHSV / RandomFlip transforms run by dataloader workers under /root/reference/BscanBased/yolo_seg_train.py:12).
 * cache: uint8 (N,H,W,3); out: uint8 (B,H,W,3); params: DEVICE array of B records.  The polygons are transformed by
 * the caller with the forward matrix (host side, dataset.py). */
typedef struct {
  int32_t src[4];          /* cache indices: top-left, top-right, bottom-left, bottom-right of the mosaic */
  float xc, yc;            /* mosaic centre on the 2W x 2H canvas */
  float minv[6];           /* canvas (u, v) = [a b c; d e f] * (x, y, 1) for output pixel (x, y) */
  float hgain, sgain, vgain;
  int32_t flip;            /* 1: mirror the output left-right */
  int32_t mosaic;          /* 0: single image src[0] at the canvas origin */
} m355_aug_params;
int m355_augment(const void* d_cache, const m355_aug_params* d_params, void* d_out, int32_t B, int32_t H, int32_t W,
                 void* stream);

/* ---- D-FINE decoder hot ops (/root/reference/D-Fine/temporal_dfine.py:160-181) ---------------
 * The reference reaches these through transformers' modeling_d_fine.py; each entry point names what it replaces.
 *
 * m355_msda_forward = multi_scale_deformable_attention_v2 (modeling_d_fine.py:150-221), called by every decoder
 * layer's cross-attention under `self.dfine.model(pixel_values=...)` (temporal_dfine.py:162).
 *   value  (B, S, heads, head_dim) fp32, S = sum of h*w over the levels, head_dim must be 32
 *   shapes_hw: HOST array [num_levels][2] = (height, width); points_per_level: HOST array, sums to P (<= 32)
 *   loc    (B, Q, heads, P, 2) fp32 (x, y); attn (B, Q, heads, P) fp32 (already soft-maxed by the caller)
 *   discrete 0: method "default" (bilinear grid_sample on 2*loc-1, zeros padding, align_corners=False)
 *            1: method "discrete" (nearest pixel of loc * (w, h) + 0.5, clamped)
 *   out    (B, Q, heads * head_dim) fp32 */
int m355_msda_forward(const float* d_value, int32_t B, int32_t S, int32_t heads, int32_t head_dim, const int32_t* shapes_hw,
                      int32_t num_levels, const float* d_loc, const float* d_attn, const int32_t* points_per_level,
                      int32_t Q, int32_t P, int32_t discrete, float* d_out, void* stream);
/* m355_msda_module_forward = the part of DFineMultiscaleDeformableAttention.forward (modeling_d_fine.py:268-311) behind its
 * two linear layers, for 4-d reference points and method "default": softmax of the attention logits over the P points,
 * sampling location = ref.xy + offset * (1 / points of the level) * ref.wh * offset_scale, then the core above.
 *   ref (B, Q, 4) cx, cy, w, h; offsets (B, Q, heads, P, 2) and logits (B, Q, heads, P): raw outputs of
 *   `sampling_offsets` / `attention_weights`; P <= 16; out (B, Q, heads * head_dim). */
int m355_msda_module_forward(const float* d_value, int32_t B, int32_t S, int32_t heads, int32_t head_dim, const int32_t* shapes_hw,
                             int32_t num_levels, const float* d_ref, const float* d_offsets, const float* d_logits,
                             const int32_t* points_per_level, int32_t Q, int32_t P, float offset_scale, float* d_out,
                             void* stream);
/* m355_dfine_decode = DFineIntegral.forward (modeling_d_fine.py:756-778) + distance2bbox (:1115-1137) [+ .clamp(0, 1)],
 * temporal_dfine.py:180-181.  dist (n, 4 * num_bins_plus1) fp32 logits; project (num_bins_plus1) fp32 = W(n) from
 * weighting_function (:1091-1112, host side: dfine.py); ref (n, 4) fp32 (cx, cy, w, h); boxes (n, 4) fp32 (cx, cy, w, h). */
int m355_dfine_decode(const float* d_dist, const float* d_project, const float* d_ref, float* d_boxes, int64_t n,
                      int32_t num_bins_plus1, float reg_scale, int32_t clamp01, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MI355YOLO_H_ */
