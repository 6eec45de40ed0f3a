// Post-processing of the YOLOv8-seg head on gfx950: batched NMS (A11) and mask assembly (A12).
// Replaces upstream utils.ops.non_max_suppression / process_mask / crop_mask reached from
// SegmentationPredictor.postprocess (call site BscanBased/yolo8_seg_predict.py:8).
//
// This file is compiled with -ffp-contract=off: the IoU / box arithmetic must round exactly like the
// oracle's float32 numpy ops so that the NMS keep-set is bit-exact for identical inputs.
#include <stdlib.h>

#include "common.h"

namespace m355 {
namespace {

constexpr int NMS_THREADS = 1024;
constexpr int MAX_KEEP = 1024;      // upper bound for max_det
constexpr float MAX_WH = 7680.0f;   // class offset (non-agnostic NMS)

struct Box { float x1, y1, x2, y2; };

__device__ __forceinline__ float iou_f32(const Box& a, const Box& b) {
  const float area_a = (a.x2 - a.x1) * (a.y2 - a.y1);
  const float area_b = (b.x2 - b.x1) * (b.y2 - b.y1);
  const float w = fmaxf(0.f, fminf(a.x2, b.x2) - fmaxf(a.x1, b.x1));
  const float h = fmaxf(0.f, fminf(a.y2, b.y2) - fmaxf(a.y1, b.y1));
  const float inter = w * h;
  const float uni = (area_a + area_b) - inter;
  return inter / uni;
}

// One 1024-thread block per image.
//  1. candidates: conf = max_c score > thr; key = (conf bits << 32) | (0xFFFFFFFF - anchor)  -> sorting
//     keys descending gives confidence descending, ties by lower anchor index first (oracle order).
//  2. bitonic sort of the keys in LDS (n padded to a power of two; <= 16384 keys = 128 KB) or, for
//     larger anchor counts, in the global workspace.
//  3. wave 0 runs greedy NMS over 64-candidate chunks: each lane owns one candidate, tests it against
//     the kept list (LDS broadcast), then the chunk is resolved in order with ballot/ffs.
__global__ __launch_bounds__(NMS_THREADS) void nms_kernel(const float* preds, int A, int nc, int nm, float conf_thr,
                                                          float iou_thr, int max_det, float* dets, int* counts,
                                                          unsigned long long* gkeys, int keys_in_lds,
                                                          int npad_max) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // layout: [kept boxes MAX_KEEP*16][kept idx MAX_KEEP*4][misc 16][keys ...]
  Box* kbox = (Box*)smem;
  int* kidx = (int*)(smem + MAX_KEEP * 16);
  int* misc = (int*)(smem + MAX_KEEP * 20);
  // keys_in_lds = number of keys the LDS buffer holds (0: none).  The candidates are counted first; the sort runs in LDS when
  // they fit and in the global workspace otherwise.  (Sizing the LDS buffer for EVERY anchor -- 128 KB at 8400 -- made this
  // block need a whole CU: beside the forward kernels of the next batch it waited for one, 148 us instead of 50.)
  unsigned long long* const lkeys = (unsigned long long*)(smem + MAX_KEEP * 20 + 16);
  unsigned long long* const wkeys = gkeys + (long)blockIdx.x * npad_max;
  const int b = blockIdx.x;
  const int tid = threadIdx.x;
  const int wd = 4 + nc + nm;
  const float* pb = preds + (long)b * A * wd;

  if (tid == 0) misc[0] = 0;
  __syncthreads();
  for (int a = tid; a < A; a += NMS_THREADS) {
    const float* p = pb + (long)a * wd + 4;
    float best = p[0];
    for (int c = 1; c < nc; ++c) best = fmaxf(best, p[c]);
    if (best > conf_thr) {
      const int slot = atomicAdd(&misc[0], 1);
      const unsigned long long key = ((unsigned long long)__float_as_uint(best) << 32) | (unsigned)(0xFFFFFFFFu - (unsigned)a);
      if (slot < keys_in_lds) lkeys[slot] = key;
      if (gkeys) wkeys[slot] = key;           // (only read when the candidates overflow the LDS buffer)
    }
  }
  __syncthreads();
  const int n = misc[0];
  unsigned long long* const keys = (n <= keys_in_lds || !gkeys) ? lkeys : wkeys;
  int npad = 1;
  while (npad < n) npad <<= 1;
  for (int i = n + tid; i < npad; i += NMS_THREADS) keys[i] = 0ull;
  __syncthreads();
  // bitonic sort, descending
  for (int k = 2; k <= npad; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < npad; i += NMS_THREADS) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const unsigned long long x = keys[i], y = keys[ixj];
          const bool desc = (i & k) == 0;
          if (desc ? (x < y) : (x > y)) {
            keys[i] = y;
            keys[ixj] = x;
          }
        }
      }
      __syncthreads();
    }
  }
  // greedy NMS by wave 0
  if (tid < 64) {
    const int lane = tid;
    int kept = 0;
    const int nchunks = (n + 63) >> 6;
    for (int c = 0; c < nchunks && kept < max_det; ++c) {
      const int i = c * 64 + lane;
      const bool valid = i < n;
      int anchor = 0;
      Box bx = {0.f, 0.f, 0.f, 0.f};
      if (valid) {
        anchor = (int)(0xFFFFFFFFu - (unsigned)(keys[i] & 0xFFFFFFFFull));
        const float* p = pb + (long)anchor * wd;
        const float cx = p[0], cy = p[1], hw = p[2] / 2.0f, hh = p[3] / 2.0f;
        int cls = 0;
        float best = p[4];
        for (int cc = 1; cc < nc; ++cc)
          if (p[4 + cc] > best) { best = p[4 + cc]; cls = cc; }
        const float off = (float)cls * MAX_WH;
        bx.x1 = (cx - hw) + off; bx.y1 = (cy - hh) + off; bx.x2 = (cx + hw) + off; bx.y2 = (cy + hh) + off;
      }
      bool alive = valid;
      for (int k = 0; k < kept; ++k) {
        const Box kb = kbox[k];
        if (alive && iou_f32(kb, bx) > iou_thr) alive = false;
      }
      unsigned long long mask = __ballot(alive);
      while (mask) {
        const int j = __ffsll((long long)mask) - 1;
        Box jb;
        jb.x1 = __shfl(bx.x1, j); jb.y1 = __shfl(bx.y1, j); jb.x2 = __shfl(bx.x2, j); jb.y2 = __shfl(bx.y2, j);
        if (lane == j) { kbox[kept] = bx; kidx[kept] = anchor; }
        ++kept;
        if (kept >= max_det) break;
        if (lane > j && alive && iou_f32(jb, bx) > iou_thr) alive = false;
        const unsigned long long above = (j == 63) ? 0ull : (~0ull << (j + 1));
        mask = __ballot(alive) & above;
      }
    }
    if (lane == 0) { misc[1] = kept; counts[b] = kept; }
  }
  __syncthreads();
  // emit rows [x1,y1,x2,y2,conf,cls,coefs...]
  const int kept = misc[1];
  const int ow = 6 + nm;
  for (int idx = tid; idx < kept * ow; idx += NMS_THREADS) {
    const int k = idx / ow, f = idx - k * ow;
    const int anchor = kidx[k];
    const float* p = pb + (long)anchor * wd;
    float v;
    if (f < 4) {
      const float cx = p[0], cy = p[1], hw = p[2] / 2.0f, hh = p[3] / 2.0f;
      v = f == 0 ? cx - hw : f == 1 ? cy - hh : f == 2 ? cx + hw : cy + hh;
    } else if (f < 6) {
      int cls = 0;
      float best = p[4];
      for (int cc = 1; cc < nc; ++cc)
        if (p[4 + cc] > best) { best = p[4 + cc]; cls = cc; }
      v = f == 4 ? best : (float)cls;
    } else {
      v = p[4 + nc + (f - 6)];
    }
    dets[((long)b * max_det + k) * ow + f] = v;
  }
}

// ---------------------------------------------------------------------------------------------
// Mask assembly (A12).  One 256-thread block per (image, 16x16 tile of the prototype grid):
//   1. the 18x18x32 prototype patch (tile + 1-cell halo, border cells replicated = the reference's index
//      clamping) is staged ONCE in LDS and shared by every detection of the image;
//   2. detections are processed 16 at a time: logits[det][cell] = coef[det] . proto[cell] is one
//      v_mfma_f32_16x16x32_f16 per 16 cells (coefficients split into fp16 hi + lo parts, two MFMAs, so the
//      fp32 coefficients lose nothing); cells outside the detection's box (scaled to the prototype grid:
//      x1 <= col < x2, y1 <= row < y2) are zeroed while the accumulators are written to LDS;
//   3. x4 bilinear upsample (align_corners=False) + threshold > 0: output X = 4q + r reads cells (q-1, q)
//      with right-cell weight 0.625 / 0.875 (r = 0, 1) and (q, q+1) with 0.125 / 0.375 (r = 2, 3); each
//      thread turns 6 cells x 2 rows into 16 output pixels = one 16-byte store.
// Detections whose box does not touch the tile's halo window get zero-filled without touching LDS.
// ---------------------------------------------------------------------------------------------
// Tile = TR x TC prototype cells -> 4 TR x 4 TC output pixels.  8 x 32 (not 16 x 16): a row segment of the tile is then 128
// output bytes = one whole cache line per 8 lanes (16 x 16 wrote 64-byte half lines: 1.67 TB/s of masks at batch 32, 22
// detections per image); same number of cells per block, same LDS.
template <int NM, int TR, int TC>
__global__ __launch_bounds__(256) void proto_masks_kernel(const float* dets, const int* counts, const half_t* protos,
                                                          int max_det, int mh, int mw, int in_h, int in_w,
                                                          uint8_t* masks, int dbg) {
  static_assert(NM == 32, "one 64-byte NHWC prototype pixel = one MFMA K step");
  constexpr int PW = TC + 2, PH = TR + 2;               // patch with its 1-cell halo
  constexpr int CELLS = PH * PW;
  constexpr int CELLS_PAD = (CELLS + 15) / 16 * 16;     // whole 16-cell MFMA column blocks
  constexpr int LG_PITCH = CELLS_PAD + 4;
  constexpr int TPR = TC / 4;                           // threads per output row (16 pixels each)
  static_assert(256 / TPR == 4 * TR, "one 16-byte store per thread covers the tile");
  __shared__ __attribute__((aligned(16))) half_t patch[CELLS_PAD * NM];
  __shared__ __attribute__((aligned(16))) float lg[16 * LG_PITCH];
  __shared__ __attribute__((aligned(16))) half_t cf_hi[16 * NM], cf_lo[16 * NM];
  __shared__ float bxs[16][4];
  __shared__ short touch_list[MAX_KEEP];     // detections whose box touches this tile's halo window
  __shared__ uint8_t touch_flag[MAX_KEEP];
  __shared__ int ntouch_s;
  const int b = blockIdx.y;
  int n = counts[b];
  if (n > max_det) n = max_det;
  if (n <= 0) return;
  const int tiles_x = (mw + TC - 1) / TC;
  const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  const float wr = (float)mw / (float)in_w, hr = (float)mh / (float)in_h;
  // halo window of this tile in prototype-grid coordinates (for the box-touch test)
  const float wx0 = (float)(tx * TC - 1), wx1 = (float)(tx * TC + TC), wy0 = (float)(ty * TR - 1),
              wy1 = (float)(ty * TR + TR);
  const int oy = tid / TPR, seg = tid - oy * TPR;
  const int Y = ty * (4 * TR) + oy, X0 = tx * (4 * TC) + seg * 16;
  const int qy = oy >> 2, ry = oy & 3;
  const int r0 = (ry < 2) ? qy : qy + 1;
  const float ly1 = (ry == 0) ? 0.625f : (ry == 1) ? 0.875f : (ry == 2) ? 0.125f : 0.375f;
  const float ly0 = 1.f - ly1;

  // 0. Which detections touch this tile?  Everything else in the tile is zeros: those stores need nothing but the boxes, so
  //    they are issued first and drain while the patch / coefficient loads of the touching detections are in flight
  //    (the kernel was bound by its load -> barrier -> MFMA -> barrier chain per 16 detections, not by the stores:
  //    68 of 107 us with the stores and the upsample switched off).
  if (tid == 0) ntouch_s = 0;
  __syncthreads();
  for (int d = tid; d < n; d += 256) {
    const float* dp = dets + ((long)b * max_det + d) * (6 + NM);
    const float x1 = dp[0] * wr, y1 = dp[1] * hr, x2 = dp[2] * wr, y2 = dp[3] * hr;
    const bool touches = !(dbg & 1) && !(wx1 < x1 || wx0 >= x2 || wy1 < y1 || wy0 >= y2);
    touch_flag[d] = touches ? 1 : 0;
    if (touches) touch_list[atomicAdd(&ntouch_s, 1)] = (short)d;   // (order is irrelevant: every detection's mask is independent)
  }
  __syncthreads();
  const int nt = ntouch_s;
  // Zero fill of the detections that do not touch the tile.  A tile nothing touches does only this.  Otherwise it is issued
  // AFTER the first chunk of touching detections: vector-memory operations retire in issue order, so a load issued behind
  // these stores (the prototype patch, the coefficients) would wait for every one of them -- fill and compute were additive
  // (76 + 48 us) with the fill first.
  auto zero_fill = [&]() __attribute__((always_inline)) {
    if (Y < in_h && !(dbg & 2)) {
      typedef unsigned uint4v __attribute__((ext_vector_type(4)));
      for (int d = 0; d < n; ++d) {
        if (touch_flag[d]) continue;
        uint8_t* mp = masks + (((long)b * max_det + d) * in_h + Y) * in_w + X0;
        if (X0 + 16 <= in_w) {
          if (dbg & 4) *(uint4v*)mp = uint4v{0u, 0u, 0u, 0u};
          else __builtin_nontemporal_store(uint4v{0u, 0u, 0u, 0u}, (uint4v*)mp);
        } else {
          for (int j = 0; j < 16 && X0 + j < in_w; ++j) mp[j] = 0;
        }
      }
    }
  };
  if (nt == 0) {
    zero_fill();
    return;
  }
  // 1. prototype patch (4 x 16-byte chunks per cell)
  const half_t* pb = protos + (long)b * mh * mw * NM;
  for (int i = tid; i < CELLS_PAD * 4; i += 256) {
    const int cell = i >> 2, ch = i & 3;
    half8 v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (cell < CELLS) {
      const int ly = cell / PW, lx = cell - ly * PW;
      int y = ty * TR - 1 + ly, x = tx * TC - 1 + lx;
      y = y < 0 ? 0 : (y >= mh ? mh - 1 : y);
      x = x < 0 ? 0 : (x >= mw ? mw - 1 : x);
      v = *(const half8*)(pb + ((long)y * mw + x) * NM + ch * 8);
    }
    *(half8*)(patch + cell * NM + ch * 8) = v;
  }
  for (int c0 = 0; c0 < nt; c0 += 16) {
    const int nd = (nt - c0 < 16) ? (nt - c0) : 16;
    __syncthreads();  // previous chunk's readers of lg / cf / bxs are done (and the patch is complete)
    // 2a. coefficients (hi + lo fp16) and scaled boxes of this chunk
    for (int i = tid; i < 16 * NM; i += 256) {
      const int d = i / NM, k = i - d * NM;
      const float c = d < nd ? dets[((long)b * max_det + touch_list[c0 + d]) * (6 + NM) + 6 + k] : 0.f;
      const half_t hi = (half_t)c;
      cf_hi[i] = hi;
      cf_lo[i] = (half_t)(c - (float)hi);
    }
    if (tid < 64) {
      const int d = tid >> 2, f = tid & 3;
      float v = 0.f;
      if (d < nd) v = dets[((long)b * max_det + touch_list[c0 + d]) * (6 + NM) + f] * ((f & 1) ? hr : wr);
      bxs[d][f] = v;
    }
    __syncthreads();
    // 2b. logits by MFMA: A = coefficients [16 dets][32], B = patch [32][16 cells]; D[det 4g+j][cell l15]
    {
      const half8 a_hi = *(const half8*)(cf_hi + l15 * NM + g * 8);
      const half8 a_lo = *(const half8*)(cf_lo + l15 * NM + g * 8);
      for (int t = wave; t < CELLS_PAD / 16; t += 4) {
        const int cell = t * 16 + l15;
        const half8 bf = *(const half8*)(patch + cell * NM + g * 8);
        float4v acc = {0.f, 0.f, 0.f, 0.f};
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi, bf, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_lo, bf, acc, 0, 0, 0);
        const int ly = cell / PW, lx = cell - ly * PW;
        int y = ty * TR - 1 + ly, x = tx * TC - 1 + lx;
        y = y < 0 ? 0 : (y >= mh ? mh - 1 : y);
        x = x < 0 ? 0 : (x >= mw ? mw - 1 : x);
        const float xf = (float)x, yf = (float)y;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int d = g * 4 + j;
          const bool in = xf >= bxs[d][0] && xf < bxs[d][2] && yf >= bxs[d][1] && yf < bxs[d][3];
          lg[d * LG_PITCH + cell] = in ? acc[j] : 0.f;
        }
      }
    }
    __syncthreads();
    // 3. upsample + threshold + store, one detection after the other (no barrier in between)
    if (Y < in_h) {
      for (int d = 0; d < nd; ++d) {
        uint8_t o[16];
        {
          const float* l0 = lg + d * LG_PITCH + r0 * PW + seg * 4;
          float cv[6];
#pragma unroll
          for (int c = 0; c < 6; ++c) cv[c] = ly0 * l0[c] + ly1 * l0[PW + c];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float a0 = cv[q], a1 = cv[q + 1], a2 = cv[q + 2];
            o[q * 4 + 0] = (0.375f * a0 + 0.625f * a1) > 0.f ? 1 : 0;
            o[q * 4 + 1] = (0.125f * a0 + 0.875f * a1) > 0.f ? 1 : 0;
            o[q * 4 + 2] = (0.875f * a1 + 0.125f * a2) > 0.f ? 1 : 0;
            o[q * 4 + 3] = (0.625f * a1 + 0.375f * a2) > 0.f ? 1 : 0;
          }
        }
        uint8_t* mp = masks + (((long)b * max_det + touch_list[c0 + d]) * in_h + Y) * in_w + X0;
        if (dbg & 2) continue;
        if (X0 + 16 <= in_w) {
          typedef unsigned uint4v __attribute__((ext_vector_type(4)));
          if (dbg & 4) *(uint4v*)mp = *(const uint4v*)o;
          else __builtin_nontemporal_store(*(const uint4v*)o, (uint4v*)mp);   // written once, never read by the GPU again
        } else {
          for (int j = 0; j < 16 && X0 + j < in_w; ++j) mp[j] = o[j];
        }
      }
    }
    if (c0 == 0) zero_fill();   // (uniform: the stores drain under the next chunk, or while the block retires)
  }
}

}  // namespace

size_t nms_workspace_bytes(int B, int A) {
  int npad = 1;
  while (npad < A) npad <<= 1;
  return (size_t)B * npad * 8;
}

int launch_nms(const float* preds, int B, int A, int nc, int nm, float conf, float iou, int max_det, float* dets,
               int* counts, void* workspace, size_t workspace_bytes, hipStream_t s) {
  if (max_det > MAX_KEEP || max_det < 1) return -1;
  int npad = 1;
  while (npad < A) npad <<= 1;
  const size_t base = MAX_KEEP * 20 + 16;
  // LDS for 4096 keys (32 KB; a power of two, the bitonic network's padding) when a global workspace exists for the
  // overflow, for every anchor otherwise
  const bool have_ws = workspace && workspace_bytes >= (size_t)B * npad * 8;
  int lds_keys = npad;
  if (have_ws && lds_keys > 4096) lds_keys = 4096;
  if (base + (size_t)lds_keys * 8 > 160 * 1024 - 256) {
    if (!have_ws) return -1;
    lds_keys = 4096;
  }
  const size_t lds = base + (size_t)lds_keys * 8;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)nms_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       160 * 1024 - 256);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  hipLaunchKernelGGL(nms_kernel, dim3(B), dim3(NMS_THREADS), lds, s, preds, A, nc, nm, conf, iou, max_det, dets,
                     counts, have_ws ? (unsigned long long*)workspace : nullptr, lds_keys, npad);
  return (int)hipGetLastError();
}

int launch_proto_masks(const float* dets, const int* counts, const half_t* protos, int B, int max_det, int nm,
                       int mh, int mw, int in_h, int in_w, uint8_t* masks, hipStream_t s) {
  if (nm != 32) return -1;
  if (in_h % mh || in_w % mw || in_w / mw != 4 || in_h / mh != 4 || in_w % 16) return -1;
  static const int dbg = getenv("M355_MASK_DBG") ? atoi(getenv("M355_MASK_DBG")) : 0;   // timing ablations: 1 zero fill only, 2 no stores
  static const int shape = getenv("M355_MASK_TILE") ? atoi(getenv("M355_MASK_TILE")) : 0;   // 1: the 16 x 16-cell tile (experiments)
  if (shape == 1) {
    const int tiles = ((mw + 15) / 16) * ((mh + 15) / 16);
    hipLaunchKernelGGL((proto_masks_kernel<32, 16, 16>), dim3(tiles, B), dim3(256), 0, s, dets, counts, protos, max_det, mh, mw,
                       in_h, in_w, masks, dbg);
  } else {
    const int tiles = ((mw + 31) / 32) * ((mh + 7) / 8);
    hipLaunchKernelGGL((proto_masks_kernel<32, 8, 32>), dim3(tiles, B), dim3(256), 0, s, dets, counts, protos, max_det, mh, mw,
                       in_h, in_w, masks, dbg);
  }
  return (int)hipGetLastError();
}

}  // namespace m355
