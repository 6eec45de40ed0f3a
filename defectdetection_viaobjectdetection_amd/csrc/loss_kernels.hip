// The mask term of the segmentation loss and its gradients (stands where v8SegmentationLoss.single_mask_loss
// + its autograd stand upstream, reached from /root/reference/BscanBased/yolo_seg_train.py:12).
//
//   slot k of image b:  pred[p] = coef[b,k,:] . proto[b,p,:]           (32 channels, p over the mh x mw prototype map)
//                       bce[p]  = BCEWithLogits(pred[p], mask[b,p] == inst[b,k])
//                       sum[b,k] = sum over the pixels p INSIDE the slot's box of bce[p]
//   L = (1 / (mh mw)) sum_{b,k} w[b,k] sum[b,k]        (w = valid / (normalised box area x number of foreground anchors))
//
// As torch ops this was one (B, K, mh*mw) fp32 GEMM and ~40 elementwise passes over tensors of that size, forward and
// backward: 1.4 ms of a 4.1 ms loss at batch 64 (rocprofv3 sequence, tools/trace_seq.py).  Only the pixels inside a box carry
// loss or gradient, and the gradient needs nothing but the forward values, so forward and backward are ONE pass here:
//   * mask_loss_slot_kernel, one block per (image, slot): walks the box, accumulates sum[b,k] and d L / d coef[b,k,:] in
//     registers, block tree reduction in a fixed order (bitwise reproducible; no float atomics);
//   * mask_loss_proto_kernel, one thread per prototype pixel: d L / d proto[b,p,:] over the slots whose box holds the pixel,
//     times the incoming gradient of L (a device scalar), stored in the prototypes' own dtype -- launched from the autograd
//     backward, so the dense map is written once, already scaled (a stored fp32 map, `* g` and `.half()` were 3 passes).
// HBM traffic is the prototype map once per kernel plus the gradient map once: ~0.2 GB at batch 64.
#include "common.h"

namespace m355 {
namespace {

constexpr int NMK = 32;   // mask coefficients = prototype channels

__device__ __forceinline__ float bce_logits(float x, float y) {   // max(x, 0) - x y + log(1 + exp(-|x|))
  return fmaxf(x, 0.f) - x * y + log1pf(expf(-fabsf(x)));
}
__device__ __forceinline__ float sigmoid_exact(float x) { return 1.0f / (1.0f + expf(-x)); }

template <bool F16>
__device__ __forceinline__ void load_proto(const void* protos, long row, float* pr) {
  if (F16) {
    const half8* q = (const half8*)((const half_t*)protos + row * NMK);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const half8 v = q[u];
#pragma unroll
      for (int j = 0; j < 8; ++j) pr[u * 8 + j] = (float)v[j];
    }
  } else {
    const float4v* q = (const float4v*)((const float*)protos + row * NMK);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float4v v = q[u];
#pragma unroll
      for (int j = 0; j < 4; ++j) pr[u * 4 + j] = v[j];
    }
  }
}

// The box test of the loss: pixel (col, row) is inside when col >= x1 && col < x2 && row >= y1 && row < y2 (float compares
// of integer-valued coordinates).  The integer ranges below are supersets; the float test decides.
struct BoxRange { int xa, xb, ya, yb; };
__device__ __forceinline__ BoxRange box_range(const float* bx, int mw, int mh) {
  BoxRange r;
  r.xa = max(0, (int)floorf(bx[0]));
  r.ya = max(0, (int)floorf(bx[1]));
  r.xb = min(mw, (int)ceilf(bx[2]) + 1);
  r.yb = min(mh, (int)ceilf(bx[3]) + 1);
  return r;
}

template <bool F16>
__global__ __launch_bounds__(256) void mask_loss_slot_kernel(const float* coef, const void* protos, const int* masks, const int* inst,
                                                             const float* boxes, const float* w, int K, int mh, int mw, float inv_hw,
                                                             float* slot_sum, float* d_coef) {
  __shared__ float red[256 * (NMK + 1)];
  const int b = blockIdx.y, k = blockIdx.x;
  const long sk = (long)b * K + k;
  const float wk = w[sk];
  float* const dco = d_coef + sk * NMK;
  if (!(wk != 0.f)) {                                   // empty slot: no loss, no gradient (also a NaN weight is not propagated)
    if (threadIdx.x < NMK) dco[threadIdx.x] = 0.f;
    if (threadIdx.x == 0) slot_sum[sk] = 0.f;
    return;
  }
  const float* const bx = boxes + sk * 4;
  const float x1 = bx[0], y1 = bx[1], x2 = bx[2], y2 = bx[3];
  const BoxRange r = box_range(bx, mw, mh);
  const int bw = r.xb - r.xa, bh = r.yb - r.ya;
  const int npx = bw > 0 && bh > 0 ? bw * bh : 0;
  const int id = inst[sk];
  float ck[NMK];
#pragma unroll
  for (int j = 0; j < NMK; ++j) ck[j] = coef[sk * NMK + j];
  float acc[NMK];
#pragma unroll
  for (int j = 0; j < NMK; ++j) acc[j] = 0.f;
  float lsum = 0.f;
  const long img = (long)b * mh * mw;
  for (int i = threadIdx.x; i < npx; i += 256) {
    const int yy = r.ya + i / bw, xx = r.xa + i % bw;
    const float col = (float)xx, row = (float)yy;
    if (!(col >= x1 && col < x2 && row >= y1 && row < y2)) continue;
    const long p = img + (long)yy * mw + xx;
    float pr[NMK];
    load_proto<F16>(protos, p, pr);
    float x = 0.f;
#pragma unroll
    for (int j = 0; j < NMK; ++j) x += ck[j] * pr[j];
    const float y = masks[p] == id ? 1.f : 0.f;
    lsum += bce_logits(x, y);
    const float g = sigmoid_exact(x) - y;
#pragma unroll
    for (int j = 0; j < NMK; ++j) acc[j] += g * pr[j];
  }
  // fixed-order block reduction of the 33 per-thread sums: columns of red[][33], halving
#pragma unroll
  for (int j = 0; j < NMK; ++j) red[threadIdx.x * (NMK + 1) + j] = acc[j];
  red[threadIdx.x * (NMK + 1) + NMK] = lsum;
  __syncthreads();
  for (int h = 128; h >= 1; h >>= 1) {
    for (int e = threadIdx.x; e < h * (NMK + 1); e += 256) {
      const int t = e / (NMK + 1), j = e - t * (NMK + 1);
      red[t * (NMK + 1) + j] += red[(t + h) * (NMK + 1) + j];
    }
    __syncthreads();
  }
  const float s = wk * inv_hw;
  if (threadIdx.x < NMK) dco[threadIdx.x] = s * red[threadIdx.x];
  if (threadIdx.x == 0) slot_sum[sk] = red[NMK];
}

template <bool F16, bool OUT16>
__global__ __launch_bounds__(256) void mask_loss_proto_kernel(const float* coef, const void* protos, const int* masks, const int* inst,
                                                              const float* boxes, const float* w, int K, int mh, int mw, float inv_hw,
                                                              const float* gscale, void* d_protos) {
  const int b = blockIdx.y;
  const int hw = mh * mw;
  const int p0 = blockIdx.x * 256;
  const int p = p0 + threadIdx.x;
  const bool live = p < hw;
  const int plast = min(p0 + 255, hw - 1);
  const int ty0 = p0 / mw, ty1 = plast / mw;              // rows this block touches
  const int yy = live ? p / mw : 0, xx = live ? p - yy * mw : 0;
  const float col = (float)xx, row = (float)yy;
  const long gp = (long)b * hw + p;
  float pr[NMK], dp[NMK];
#pragma unroll
  for (int j = 0; j < NMK; ++j) dp[j] = 0.f;
  int m = 0;
  if (live) {
    load_proto<F16>(protos, gp, pr);
    m = masks[gp];
  } else {
#pragma unroll
    for (int j = 0; j < NMK; ++j) pr[j] = 0.f;
  }
  for (int k = 0; k < K; ++k) {
    const long sk = (long)b * K + k;                    // block-uniform: scalar loads, uniform branches
    const float wk = w[sk];
    if (!(wk != 0.f)) continue;
    const float* const bx = boxes + sk * 4;
    const float x1 = bx[0], y1 = bx[1], x2 = bx[2], y2 = bx[3];
    if (!((float)ty1 >= y1 && (float)ty0 < y2)) continue;   // the box misses every row of this block
    const float* const ck = coef + sk * NMK;
    if (live && col >= x1 && col < x2 && row >= y1 && row < y2) {
      float x = 0.f;
#pragma unroll
      for (int j = 0; j < NMK; ++j) x += ck[j] * pr[j];
      const float y = m == inst[sk] ? 1.f : 0.f;
      const float g = (sigmoid_exact(x) - y) * (wk * inv_hw);
#pragma unroll
      for (int j = 0; j < NMK; ++j) dp[j] += g * ck[j];
    }
  }
  if (live) {
    const float gs = gscale ? *gscale : 1.0f;             // the incoming gradient of L (a device scalar): applied here, not in a pass of its own
    if (OUT16) {
      half8* o = (half8*)((half_t*)d_protos + gp * NMK);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        half8 h;
#pragma unroll
        for (int j = 0; j < 8; ++j) h[j] = m355_to_half(dp[u * 8 + j] * gs);
        o[u] = h;
      }
    } else {
      float4v* o = (float4v*)((float*)d_protos + gp * NMK);
#pragma unroll
      for (int u = 0; u < 8; ++u) o[u] = float4v{dp[u * 4] * gs, dp[u * 4 + 1] * gs, dp[u * 4 + 2] * gs, dp[u * 4 + 3] * gs};
    }
  }
}
}  // namespace

// ---------------------------------------------------------------------------------------------------------
// Box (CIoU) and DFL terms on the foreground slots, value and gradient w.r.t. the 4 x 16 distribution logits of a slot in one
// pass (upstream: BboxLoss.forward + bbox_iou(CIoU=True) + DFLoss and their autograd).  As torch ops on (B, K) tensors this
// was ~60 forward and ~170 backward launches of a few microseconds each on 1 280 elements.  One thread per slot:
//   ltrb_i = sum_j softmax(z_i)_j j;  pred = (ax - l, ay - t, ax + r, ay + b);  box = w (1 - CIoU(pred, target));
//   dfl = w mean_i CE(z_i; lo_i, lo_i + 1 with weights (lo_i + 1 - d_i), (d_i - lo_i)),  d_i = clamp(target distance, 0, 14.99).
// dCIoU/d(l,t,r,b) by forward-mode differentiation (a value with four tangents) of the very expression loss.ciou evaluates,
// alpha held constant as there (torch.no_grad).
// ---------------------------------------------------------------------------------------------------------
namespace {
constexpr int REGM = 16;
constexpr float CEPS = 1e-7f;

struct D4 {
  float v, d[4];
};
__device__ __forceinline__ D4 dc(float c) { return D4{c, {0.f, 0.f, 0.f, 0.f}}; }
__device__ __forceinline__ D4 operator+(D4 a, D4 b) { return D4{a.v + b.v, {a.d[0] + b.d[0], a.d[1] + b.d[1], a.d[2] + b.d[2], a.d[3] + b.d[3]}}; }
__device__ __forceinline__ D4 operator-(D4 a, D4 b) { return D4{a.v - b.v, {a.d[0] - b.d[0], a.d[1] - b.d[1], a.d[2] - b.d[2], a.d[3] - b.d[3]}}; }
__device__ __forceinline__ D4 operator*(D4 a, D4 b) {
  return D4{a.v * b.v, {a.d[0] * b.v + a.v * b.d[0], a.d[1] * b.v + a.v * b.d[1], a.d[2] * b.v + a.v * b.d[2], a.d[3] * b.v + a.v * b.d[3]}};
}
__device__ __forceinline__ D4 operator/(D4 a, D4 b) {
  const float q = a.v / b.v, r = 1.0f / b.v;
  return D4{q, {(a.d[0] - q * b.d[0]) * r, (a.d[1] - q * b.d[1]) * r, (a.d[2] - q * b.d[2]) * r, (a.d[3] - q * b.d[3]) * r}};
}
__device__ __forceinline__ D4 dscale(D4 a, float c) { return D4{a.v * c, {a.d[0] * c, a.d[1] * c, a.d[2] * c, a.d[3] * c}}; }
__device__ __forceinline__ D4 davg(D4 a, D4 b) { return dscale(a + b, 0.5f); }
// torch.minimum / maximum: the gradient goes to the selected operand, half to each on a tie
__device__ __forceinline__ D4 dmin(D4 a, D4 b) { return a.v < b.v ? a : (b.v < a.v ? b : davg(a, b)); }
__device__ __forceinline__ D4 dmax(D4 a, D4 b) { return a.v > b.v ? a : (b.v > a.v ? b : davg(a, b)); }
__device__ __forceinline__ D4 dclamp0(D4 a) { return a.v >= 0.f ? a : dc(0.f); }       // clamp_min(0): gradient where x >= 0
__device__ __forceinline__ D4 datan(D4 a) {
  const float r = 1.0f / (1.0f + a.v * a.v);
  return D4{atanf(a.v), {a.d[0] * r, a.d[1] * r, a.d[2] * r, a.d[3] * r}};
}

__device__ __forceinline__ D4 ciou_d4(D4 ax1, D4 ay1, D4 ax2, D4 ay2, float bx1, float by1, float bx2, float by2) {
  const D4 aw = ax2 - ax1, ah = ay2 - ay1 + dc(CEPS);
  const float bw = bx2 - bx1, bh = by2 - by1 + CEPS;
  const D4 iw = dclamp0(dmin(ax2, dc(bx2)) - dmax(ax1, dc(bx1)));
  const D4 ih = dclamp0(dmin(ay2, dc(by2)) - dmax(ay1, dc(by1)));
  const D4 inter = iw * ih;
  const D4 iou = inter / (aw * ah + dc(bw * bh) - inter + dc(CEPS));
  const D4 cw = dmax(ax2, dc(bx2)) - dmin(ax1, dc(bx1));
  const D4 ch = dmax(ay2, dc(by2)) - dmin(ay1, dc(by1));
  const D4 diag2 = cw * cw + ch * ch + dc(CEPS);
  const D4 ex = dc(bx1 + bx2) - ax1 - ax2, ey = dc(by1 + by2) - ay1 - ay2;
  const D4 centre2 = dscale(ex * ex + ey * ey, 0.25f);
  const D4 da = dc(atanf(bw / bh)) - datan(aw / ah);
  const D4 v = dscale(da * da, (float)(4.0 / (3.14159265358979323846 * 3.14159265358979323846)));
  const float alpha = v.v / (v.v - iou.v + (1.0f + CEPS));
  return iou - (centre2 / diag2 + dscale(v, alpha));
}

__global__ __launch_bounds__(64) void box_loss_kernel(const float* logits, const float* anchors, const float* targets, const float* weights,
                                                      long n, float* box_term, float* dfl_term, float* d_box, float* d_dfl) {
  const long s = (long)blockIdx.x * 64 + threadIdx.x;
  if (s >= n) return;
  const float w = weights[s];
  float* const gb = d_box + s * 4 * REGM;
  float* const gd = d_dfl + s * 4 * REGM;
  if (!(w != 0.f)) {
    for (int j = 0; j < 4 * REGM; ++j) { gb[j] = 0.f; gd[j] = 0.f; }
    box_term[s] = 0.f;
    dfl_term[s] = 0.f;
    return;
  }
  const float ax = anchors[s * 2], ay = anchors[s * 2 + 1];
  const float tx1 = targets[s * 4], ty1 = targets[s * 4 + 1], tx2 = targets[s * 4 + 2], ty2 = targets[s * 4 + 3];
  const float tdist[4] = {ax - tx1, ay - ty1, tx2 - ax, ty2 - ay};
  float p[4][REGM], E[4], ce[4], wl[4];
  int lo[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float z[REGM];
    float m = -INFINITY;
#pragma unroll
    for (int j = 0; j < REGM; ++j) { z[j] = logits[(s * 4 + i) * REGM + j]; m = fmaxf(m, z[j]); }
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < REGM; ++j) { p[i][j] = expf(z[j] - m); sum += p[i][j]; }
    const float inv = 1.0f / sum, lse = m + logf(sum);
    float e = 0.f;
#pragma unroll
    for (int j = 0; j < REGM; ++j) { p[i][j] *= inv; e += p[i][j] * (float)j; }
    E[i] = e;
    const float dist = fminf(fmaxf(tdist[i], 0.f), (float)(REGM - 1) - 0.01f);
    lo[i] = (int)dist;
    wl[i] = (float)(lo[i] + 1) - dist;
    float zlo = 0.f, zhi = 0.f;
#pragma unroll
    for (int j = 0; j < REGM; ++j) { zlo = j == lo[i] ? z[j] : zlo; zhi = j == lo[i] + 1 ? z[j] : zhi; }
    ce[i] = (lse - zlo) * wl[i] + (lse - zhi) * (dist - (float)lo[i]);
  }
  const D4 x1 = D4{ax - E[0], {-1.f, 0.f, 0.f, 0.f}}, y1 = D4{ay - E[1], {0.f, -1.f, 0.f, 0.f}};
  const D4 x2 = D4{ax + E[2], {0.f, 0.f, 1.f, 0.f}}, y2 = D4{ay + E[3], {0.f, 0.f, 0.f, 1.f}};
  const D4 c = ciou_d4(x1, y1, x2, y2, tx1, ty1, tx2, ty2);
  box_term[s] = (1.0f - c.v) * w;
  dfl_term[s] = ((ce[0] + ce[1] + ce[2] + ce[3]) * 0.25f) * w;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float gbox = -c.d[i] * w;
    const float wh = 1.0f - wl[i];
#pragma unroll
    for (int j = 0; j < REGM; ++j) {
      gb[i * REGM + j] = gbox * p[i][j] * ((float)j - E[i]);
      gd[i * REGM + j] = 0.25f * w * (p[i][j] - (j == lo[i] ? wl[i] : 0.f) - (j == lo[i] + 1 ? wh : 0.f));
    }
  }
}

// Boxes and class scores of ALL anchors for the assignment (no gradient): rows of the raw head map (rw = 64 + nc + nm floats,
// 4-byte aligned rows) -> xyxy in pixels ((anchor -/+ expectation) * stride) and sigmoid(class logits).  64 rows per block
// come in through LDS with coalesced loads (row pitch rw + 1 floats when rw is even: conflict-free column walks);
// thread (row, side) owns 16 logits.
__global__ __launch_bounds__(256) void dfl_decode_kernel(const float* raw, long rows, int A, int rw, int nc, const float* anchors,
                                                         const float* strides, float* boxes, float* scores) {
  extern __shared__ float tile[];
  const int pitch = rw | 1;
  const long r0 = (long)blockIdx.x * 64;
  const int nr = rows - r0 >= 64 ? 64 : (int)(rows - r0);
  const float* const src = raw + r0 * rw;
  for (int i = threadIdx.x; i < nr * rw; i += 256) {
    const int r = i / rw, c = i - r * rw;
    tile[r * pitch + c] = src[i];
  }
  __syncthreads();
  const int r = threadIdx.x >> 2, side = threadIdx.x & 3;
  if (r < nr) {
    const float* z = tile + r * pitch + side * REGM;
    float m = -INFINITY;
#pragma unroll
    for (int j = 0; j < REGM; ++j) m = fmaxf(m, z[j]);
    float t[REGM], sum = 0.f;
#pragma unroll
    for (int j = 0; j < REGM; ++j) {
      t[j] = expf(z[j] - m);
      sum += t[j];
    }
    // softmax, then the expectation, as `(softmax * bins).sum()`: every p_j = t_j / sum is rounded before its product
    const float inv = 1.0f / sum;
    float ex = 0.f;
#pragma unroll
    for (int j = 0; j < REGM; ++j) ex += (t[j] * inv) * (float)j;
    const long row = r0 + r;
    const int a = (int)(row % A);
    const float anc = anchors[a * 2 + (side & 1)];
    const float v = side < 2 ? anc - ex : anc + ex;
    boxes[row * 4 + side] = v * strides[a];
  }
  for (int i = threadIdx.x; i < nr * nc; i += 256) {
    const int rr = i / nc, c = i - rr * nc;
    scores[(r0 + rr) * nc + c] = sigmoid_exact(tile[rr * pitch + 4 * REGM + c]);
  }
}
// ---------------------------------------------------------------------------------------------------------------------------
// Task-aligned assignment (upstream TaskAlignedAssigner, reached from
// /root/reference/BscanBased/yolo_seg_train.py:12): loss.assign_targets as two kernels instead of ~40 torch launches on
// (B, G, A) tensors (round-3 verdict: "task-aligned assignment + class BCE as one or two kernels").
//   tal_topk_kernel     one block per (image, ground truth): candidate test (anchor centre strictly inside the box), CIoU with the
//                       predicted box (the expression loss.ciou evaluates, fp32), metric = score^0.5 * CIoU^6, the ten largest
//                       metrics (ties: lower anchor index) -> per (b, g) ten (anchor, metric, overlap) records; a picked anchor that
//                       is no candidate is dropped (top & cand)
//   tal_resolve_kernel  one block per image: an anchor claimed by several ground truths goes to the one with the highest CIoU over
//                       ALL ground truths (first maximum), the per-truth maxima of metric and overlap over the final positives give
//                       norm = metric * max_overlap / (max_metric + eps); writes target boxes / scores / foreground / index for the
//                       positives only (the outputs are zero-filled by the caller)
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int TAL_K = 10;

__device__ __forceinline__ float tal_ciou(const float* g, const float* b) {   // loss.ciou(gt, pred), complete
  const float eps = 1e-7f;
  const float aw = g[2] - g[0], ah = g[3] - g[1] + eps;
  const float bw = b[2] - b[0], bh = b[3] - b[1] + eps;
  const float iw = fmaxf(fminf(g[2], b[2]) - fmaxf(g[0], b[0]), 0.f);
  const float ih = fmaxf(fminf(g[3], b[3]) - fmaxf(g[1], b[1]), 0.f);
  const float inter = iw * ih;
  const float iou = inter / (aw * ah + bw * bh - inter + eps);
  const float cw = fmaxf(g[2], b[2]) - fminf(g[0], b[0]);
  const float ch = fmaxf(g[3], b[3]) - fminf(g[1], b[1]);
  const float diag2 = cw * cw + ch * ch + eps;
  const float dx = b[0] + b[2] - g[0] - g[2], dy = b[1] + b[3] - g[1] - g[3];
  const float centre2 = (dx * dx + dy * dy) * 0.25f;
  const float da = atanf(bw / bh) - atanf(aw / ah);
  const float v = 0.40528473456935109f * da * da;                             // 4 / pi^2
  const float alpha = v / (v - iou + (1.0f + eps));
  return iou - (centre2 / diag2 + v * alpha);
}
__device__ __forceinline__ bool tal_cand(const float* g, float ax, float ay) {
  const float d = fminf(fminf(ax - g[0], ay - g[1]), fminf(g[2] - ax, g[3] - ay));
  return d > 1e-9f;
}
__device__ __forceinline__ float tal_metric(float score, float ov) {         // score^0.5 * ov^6 (ov >= 0): torch.pow(x, 0.5) is sqrt,
  return sqrtf(score) * powf(ov, 6.0f);                                       // torch.pow(x, 6.0) the pow function -- the ORDER of the metrics decides
}

#pragma clang fp contract(off)
__global__ __launch_bounds__(256) void tal_topk_kernel(const float* scores, const float* boxes, const float* anchors_px, const int* gt_cls,
                                                      const float* gt_boxes, const unsigned char* gt_valid, int A, int G, int nc,
                                                      int* top_idx, float* top_metric, float* top_overlap) {
  extern __shared__ float sm[];          // A metrics + A overlaps
  float* met = sm;
  float* ovl = sm + A;
  __shared__ float rv[256];
  __shared__ int ri[256];
  const int g = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const long bg = (long)b * G + g;
  const float gb[4] = {gt_boxes[bg * 4], gt_boxes[bg * 4 + 1], gt_boxes[bg * 4 + 2], gt_boxes[bg * 4 + 3]};
  const bool valid = gt_valid[bg] != 0;
  int c = gt_cls[bg];
  c = c < 0 ? 0 : (c > nc - 1 ? nc - 1 : c);
  for (int a = tid; a < A; a += 256) {
    const bool cand = valid && tal_cand(gb, anchors_px[2 * a], anchors_px[2 * a + 1]);
    float ov = 0.f, m = 0.f;
    if (cand) {
      const float* pb = boxes + ((long)b * A + a) * 4;
      const float bx[4] = {pb[0], pb[1], pb[2], pb[3]};
      ov = fmaxf(tal_ciou(gb, bx), 0.f);
      m = tal_metric(scores[((long)b * A + a) * nc + c], ov);
    }
    ovl[a] = ov;
    met[a] = cand ? m : -1.0f;           // non-candidates can be picked by upstream's topk only among zero ties, and are dropped again
  }
  __syncthreads();
  for (int k = 0; k < TAL_K; ++k) {
    float best = -2.0f;
    int bi = 0x7fffffff;
    for (int a = tid; a < A; a += 256) {
      const float m = met[a];
      if (m > best) { best = m; bi = a; }   // ascending a: the first maximum of this thread's anchors
    }
    rv[tid] = best; ri[tid] = bi;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if (tid < s) {
        const float o = rv[tid + s];
        const int oi = ri[tid + s];
        if (o > rv[tid] || (o == rv[tid] && oi < ri[tid])) { rv[tid] = o; ri[tid] = oi; }
      }
      __syncthreads();
    }
    if (tid == 0) {
      const int a = ri[0];
      const bool pos = rv[0] >= 0.f && a < A;          // a candidate (metric may be 0: upstream keeps it when it is among the top ten)
      top_idx[bg * TAL_K + k] = pos ? a : -1;
      top_metric[bg * TAL_K + k] = pos ? rv[0] : 0.f;
      top_overlap[bg * TAL_K + k] = pos ? ovl[a] : 0.f;
      if (a < A) met[a] = -3.0f;
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void tal_resolve_kernel(const float* scores, const float* boxes, const float* anchors_px, const int* gt_cls,
                                                         const float* gt_boxes, const unsigned char* gt_valid, int A, int G, int nc,
                                                         const int* top_idx, const float* top_metric, const float* top_overlap,
                                                         float* t_boxes, float* t_scores, unsigned char* fg, long* gt_idx) {
  extern __shared__ int smi[];            // n anchors | n final g | n metric | n overlap | G max metric | G max overlap
  const int b = blockIdx.x, tid = threadIdx.x, n = G * TAL_K;
  int* e_a = smi;
  int* e_g = smi + n;
  float* e_m = (float*)(smi + 2 * n);
  float* e_o = (float*)(smi + 3 * n);
  float* mx_m = (float*)(smi + 4 * n);
  float* mx_o = mx_m + G;
  for (int e = tid; e < n; e += 256) e_a[e] = top_idx[(long)b * n + e];
  for (int g = tid; g < G; g += 256) { mx_m[g] = 0.f; mx_o[g] = 0.f; }
  __syncthreads();
  for (int e = tid; e < n; e += 256) {
    const int a = e_a[e];
    int fgq = -1;
    float m = 0.f, o = 0.f;
    if (a >= 0) {
      int claims = 0, first = e;
      for (int f = 0; f < n; ++f)
        if (e_a[f] == a) { ++claims; first = f < first ? f : first; }
      if (claims == 1) {
        fgq = e / TAL_K; m = top_metric[(long)b * n + e]; o = top_overlap[(long)b * n + e];
      } else if (first == e) {            // one record per multiply-claimed anchor: the ground truth with the highest overlap of ALL
        const float ax = anchors_px[2 * a], ay = anchors_px[2 * a + 1];
        const float* pb = boxes + ((long)b * A + a) * 4;
        const float bx[4] = {pb[0], pb[1], pb[2], pb[3]};
        float best = -1.f;
        for (int g = 0; g < G; ++g) {
          const float* gp = gt_boxes + ((long)b * G + g) * 4;
          const float gb[4] = {gp[0], gp[1], gp[2], gp[3]};
          const bool cand = gt_valid[(long)b * G + g] != 0 && tal_cand(gb, ax, ay);
          const float ov = cand ? fmaxf(tal_ciou(gb, bx), 0.f) : 0.f;
          if (ov > best) { best = ov; fgq = g; }
        }
        int c = gt_cls[(long)b * G + fgq];
        c = c < 0 ? 0 : (c > nc - 1 ? nc - 1 : c);
        o = best;
        const float* gp = gt_boxes + ((long)b * G + fgq) * 4;
        const float gb[4] = {gp[0], gp[1], gp[2], gp[3]};
        const bool cand = gt_valid[(long)b * G + fgq] != 0 && tal_cand(gb, ax, ay);
        m = cand ? tal_metric(scores[((long)b * A + a) * nc + c], o) : 0.f;
      }
    }
    e_g[e] = fgq; e_m[e] = m; e_o[e] = o;
  }
  __syncthreads();
  for (int g = tid; g < G; g += 256) {   // maxima over the truth's final positives (fixed order)
    float mm = 0.f, mo = 0.f;
    for (int e = 0; e < n; ++e)
      if (e_g[e] == g) { mm = fmaxf(mm, e_m[e]); mo = fmaxf(mo, e_o[e]); }
    mx_m[g] = mm; mx_o[g] = mo;
  }
  __syncthreads();
  for (int e = tid; e < n; e += 256) {
    const int g = e_g[e];
    if (g < 0) continue;
    const int a = e_a[e];
    const long ba = (long)b * A + a;
    const float norm = e_m[e] * mx_o[g] / (mx_m[g] + 1e-9f);
    int c = gt_cls[(long)b * G + g];
    c = c < 0 ? 0 : (c > nc - 1 ? nc - 1 : c);
    const float* gp = gt_boxes + ((long)b * G + g) * 4;
    t_boxes[ba * 4] = gp[0]; t_boxes[ba * 4 + 1] = gp[1]; t_boxes[ba * 4 + 2] = gp[2]; t_boxes[ba * 4 + 3] = gp[3];
    t_scores[ba * nc + c] = norm;
    fg[ba] = 1;
    gt_idx[ba] = g;
  }
}

}  // namespace

int launch_box_loss(const float* logits, const float* anchors, const float* targets, const float* weights, long n, float* box_term,
                    float* dfl_term, float* d_box, float* d_dfl, hipStream_t s) {
  if (!logits || !anchors || !targets || !weights || !box_term || !dfl_term || !d_box || !d_dfl || n < 1) return -1;
  hipLaunchKernelGGL(box_loss_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, logits, anchors, targets, weights, n, box_term,
                     dfl_term, d_box, d_dfl);
  return (int)hipGetLastError();
}

int launch_dfl_decode(const float* raw, long rows, int A, int rw, int nc, const float* anchors, const float* strides, float* boxes,
                      float* scores, hipStream_t s) {
  if (!raw || !anchors || !strides || !boxes || !scores || rows < 1 || A < 1 || nc < 1 || rw < 4 * REGM + nc) return -1;
  const size_t lds = (size_t)64 * (rw | 1) * sizeof(float);
  if (lds > 64 * 1024) return -1;
  hipLaunchKernelGGL(dfl_decode_kernel, dim3((unsigned)((rows + 63) / 64)), dim3(256), lds, s, raw, rows, A, rw, nc, anchors, strides,
                     boxes, scores);
  return (int)hipGetLastError();
}

int launch_mask_loss(const float* coef, const void* protos, int protos_f16, const int* masks, const int* inst, const float* boxes,
                     const float* w, int B, int K, int mh, int mw, float* slot_sum, float* d_coef, void* d_protos, int d_protos_f16,
                     const float* gscale, hipStream_t s) {
  if (!coef || !protos || !masks || !inst || !boxes || !w || (!slot_sum != !d_coef) || (!slot_sum && !d_protos)) return -1;
  if (B < 1 || K < 1 || mh < 1 || mw < 1 || K > 65535 || B > 65535 || (long)mh * mw > (1L << 30)) return -1;
  const float inv_hw = 1.0f / (float)((long)mh * mw);
  const dim3 gs(K, B), gp((mh * mw + 255) / 256, B);
  if (slot_sum) {
    if (protos_f16)
      hipLaunchKernelGGL(mask_loss_slot_kernel<true>, gs, dim3(256), 0, s, coef, protos, masks, inst, boxes, w, K, mh, mw, inv_hw, slot_sum,
                         d_coef);
    else
      hipLaunchKernelGGL(mask_loss_slot_kernel<false>, gs, dim3(256), 0, s, coef, protos, masks, inst, boxes, w, K, mh, mw, inv_hw, slot_sum,
                         d_coef);
  }
  if (d_protos) {
#define M355_PROTO_LAUNCH(F, O) \
  hipLaunchKernelGGL((mask_loss_proto_kernel<F, O>), gp, dim3(256), 0, s, coef, protos, masks, inst, boxes, w, K, mh, mw, inv_hw, gscale, d_protos)
    if (protos_f16 && d_protos_f16) M355_PROTO_LAUNCH(true, true);
    else if (protos_f16) M355_PROTO_LAUNCH(true, false);
    else if (d_protos_f16) M355_PROTO_LAUNCH(false, true);
    else M355_PROTO_LAUNCH(false, false);
#undef M355_PROTO_LAUNCH
  }
  return (int)hipGetLastError();
}

// Task-aligned assignment on the device.  scores (B,A,nc), boxes (B,A,4) xyxy px, anchors_px (A,2), gt_cls (B,G) int32, gt_boxes
// (B,G,4), gt_valid (B,G) bytes; ws: (B * G * 10) ints + 2 x (B * G * 10) floats; the four outputs must be zero on entry.
int launch_tal_assign(const float* scores, const float* boxes, const float* anchors_px, const int* gt_cls, const float* gt_boxes,
                      const unsigned char* gt_valid, int B, int A, int G, int nc, void* ws, float* t_boxes, float* t_scores,
                      unsigned char* fg, long* gt_idx, hipStream_t s) {
  if (B < 1 || A < 1 || G < 1 || nc < 1 || !ws) return -1;
  const long n = (long)B * G * TAL_K;
  int* top_idx = (int*)ws;
  float* top_metric = (float*)ws + n;
  float* top_overlap = (float*)ws + 2 * n;
  const size_t lds1 = (size_t)2 * A * sizeof(float), lds2 = (size_t)(4 * G * TAL_K + 2 * G) * sizeof(int);
  if (lds1 > 150 * 1024 || lds2 > 150 * 1024) return -1;
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute((const void*)tal_topk_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess) return -2;
    if (hipFuncSetAttribute((const void*)tal_resolve_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess) return -2;
    attr = true;
  }
  hipLaunchKernelGGL(tal_topk_kernel, dim3(G, B), dim3(256), lds1, s, scores, boxes, anchors_px, gt_cls, gt_boxes, gt_valid, A, G, nc, top_idx,
                     top_metric, top_overlap);
  hipLaunchKernelGGL(tal_resolve_kernel, dim3(B), dim3(256), lds2, s, scores, boxes, anchors_px, gt_cls, gt_boxes, gt_valid, A, G, nc, top_idx,
                     top_metric, top_overlap, t_boxes, t_scores, fg, gt_idx);
  return (int)hipGetLastError();
}

}  // namespace m355
