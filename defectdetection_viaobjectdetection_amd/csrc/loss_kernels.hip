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
//   * mask_loss_proto_kernel, one thread per prototype pixel: d L / d proto[b,p,:] over the slots whose box holds the pixel.
// HBM traffic is the prototype map once per kernel plus the dense gradient map (fp32): ~0.3 GB at batch 64.
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

template <bool F16>
__global__ __launch_bounds__(256) void mask_loss_proto_kernel(const float* coef, const void* protos, const int* masks, const int* inst,
                                                              const float* boxes, const float* w, int K, int mh, int mw, float inv_hw,
                                                              float* d_protos) {
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
    float4v* o = (float4v*)(d_protos + gp * NMK);
#pragma unroll
    for (int u = 0; u < 8; ++u) o[u] = float4v{dp[u * 4], dp[u * 4 + 1], dp[u * 4 + 2], dp[u * 4 + 3]};
  }
}
}  // namespace

int launch_mask_loss(const float* coef, const void* protos, int protos_f16, const int* masks, const int* inst, const float* boxes,
                     const float* w, int B, int K, int mh, int mw, float* slot_sum, float* d_coef, float* d_protos, hipStream_t s) {
  if (!coef || !protos || !masks || !inst || !boxes || !w || !slot_sum || !d_coef || !d_protos) return -1;
  if (B < 1 || K < 1 || mh < 1 || mw < 1 || K > 65535 || B > 65535 || (long)mh * mw > (1L << 30)) return -1;
  const float inv_hw = 1.0f / (float)((long)mh * mw);
  const dim3 gs(K, B), gp((mh * mw + 255) / 256, B);
  if (protos_f16) {
    hipLaunchKernelGGL(mask_loss_slot_kernel<true>, gs, dim3(256), 0, s, coef, protos, masks, inst, boxes, w, K, mh, mw, inv_hw, slot_sum,
                       d_coef);
    hipLaunchKernelGGL(mask_loss_proto_kernel<true>, gp, dim3(256), 0, s, coef, protos, masks, inst, boxes, w, K, mh, mw, inv_hw,
                       d_protos);
  } else {
    hipLaunchKernelGGL(mask_loss_slot_kernel<false>, gs, dim3(256), 0, s, coef, protos, masks, inst, boxes, w, K, mh, mw, inv_hw, slot_sum,
                       d_coef);
    hipLaunchKernelGGL(mask_loss_proto_kernel<false>, gp, dim3(256), 0, s, coef, protos, masks, inst, boxes, w, K, mh, mw, inv_hw,
                       d_protos);
  }
  return (int)hipGetLastError();
}

}  // namespace m355
