// The three output convs of one head level (cv2.x.2 64 -> 64 box logits, cv3.x.2 c3 -> nc, cv4.x.2 32 -> nm: a block-diagonal
// 1x1 over the 224-channel branch tensor) AND the decode of their rows (DFL expectation, dist2bbox, x stride, class sigmoid,
// coefficient copy) in one launch (gfx950).  Replaces, on the predict path (raw head maps not requested), the launch
// conv_igemm<128x128,k1> per level + head_decode_kernel: at batch 32 those were 59 + 22 + 17 + 41 = 139 us for a step that moves
// 120 MB in (fp16 branch tensors) and 40 MB out (prediction rows): the im2col kernel writes 97 fp32 raw values per anchor
// (104 MB) that the decode reads back.  Upstream: Detect / Segment forward, SURVEY.md A9 / A10 (ultralytics nn/modules/head.py
// as reached from BscanBased/yolo8_seg_predict.py:8).
//
// The arithmetic after the MFMA is head_decode_kernel's, operation for operation (acc + bias; max, __expf, the two running sums
// in bin order, IEEE division; (x1 + x2) * 0.5f * stride; 1 / (1 + __expf(-z))).  The dot products run on another MFMA shape
// (32x32x16 here, 16x16x32 there) and still come out the same: the prediction rows are bit-identical to the two-launch path on
// every shape of tests/test_engine_gpu.py::test_head_levels_as_conv_plus_decode_launches (nc = 1, 3, 20).
//
// Block = 8 waves, tile = 128 consecutive pixels of the level's flat (image, y, x) axis = 57 344 contiguous bytes, staged by
// LDS-DMA in 56 pieces of 1 KiB into one of two buffers (the next tile streams in under this one); pixel rows of 448 bytes,
// 16-byte chunk index XOR-ed with (pixel >> 2) & 3 on the source side and on the read (conflict-free for the four service groups
// of ds_read_b128).  Wave q < 4: the box block of pixel block q -- 2 x 4 MFMAs, weights in 32 VGPRs; with the plain row order
// lane-half h of channel block b holds the 16 bins of side 2 b + h of its pixel, so the softmax expectation is lane-local and
// lane h = 0 ends with (cx, w), lane h = 1 with (cy, h).  Wave q + 4: class block (8 MFMAs over the c3 = 128 class-branch
// channels, rows >= nc are zero) and coefficient block (2 MFMAs).  Rows are assembled in LDS ([32][4 + nc + nm] floats per
// pixel block) and leave as 16-byte stores of whole rows.  HBM-bound by design: 596 bytes per anchor.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include "common.h"

namespace m355 {
namespace {

typedef float float16v __attribute__((ext_vector_type(16)));

constexpr int TP = 128, ROWB = 448, NWAVES = 8;      // (28 chunks of 16 bytes per pixel row)
constexpr int TILE_BYTES = TP * ROWB;                // 57344
constexpr int NPIECES = TILE_BYTES / 1024;           // 56
constexpr int P_IT = NPIECES / NWAVES;               // 7
constexpr int WO_MAX = 4 + 32 + 32;
constexpr int STG_OFF = 2 * TILE_BYTES;              // 4 pixel blocks x 32 rows x wo floats
constexpr int STG_BLOCK = 32 * WO_MAX * 4 + 256;     // (per pixel block)
constexpr int BIAS_OFF = STG_OFF + 4 * STG_BLOCK;    // 64 + nc + nm floats
constexpr int ROWT_OFF = BIAS_OFF + 512;             // 128 row bases (float index into preds, as long)
constexpr int LDS_BYTES = ROWT_OFF + TP * 8;         // 151 040

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, int voff, int soff, char* lds) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds, 16, voff, soff, 0, 0);
}
__device__ __forceinline__ int lane_id() {
  int ln;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
  return ln;
}

// the DFL expectation of one side: head_decode_kernel's loop over the 16 bins, bias added first
__device__ __forceinline__ float dfl_side(const float16v& acc, const float* bias16) {
  float v[16];
#pragma unroll
  for (int q4 = 0; q4 < 4; ++q4) {
    const float4v u = *(const float4v*)(bias16 + 4 * q4);
#pragma unroll
    for (int j = 0; j < 4; ++j) v[4 * q4 + j] = acc[4 * q4 + j] + u[j];
  }
  float mx = v[0];
#pragma unroll
  for (int j = 1; j < 16; ++j) mx = fmaxf(mx, v[j]);
  float se = 0.f, sw = 0.f;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const float e = __expf(v[j] - mx);
    se += e;
    sw += e * (float)j;
  }
  return sw / se;
}

__global__ __launch_bounds__(512, 2) void head_tail_kernel(const HeadTailArgs a, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 31, h = lane >> 5;
  const int q = wave & 3, role = wave >> 2;          // pixel block, 0: box, 1: class + coefficients
  const int nc = a.nc, nm = a.nm, wo = 4 + nc + nm;
  const int nwg = gridDim.x;

  for (int i = tid; i < 64 + nc + nm; i += 64 * NWAVES) ((float*)(smem + BIAS_OFF))[i] = a.bias[i];

  // ---- this wave's weight fragments: box 2 x 4 (fragments 0-7) or class 8 + coefficients 2 (fragments 8-17)
  half8 wv[10];
  {
    const half_t* wp = a.wf + (long)(role ? 8 : 0) * 512 + lane * 8;
#pragma unroll
    for (int s = 0; s < 10; ++s) wv[s] = (role || s < 8) ? *(const half8*)(wp + 512 * s) : (half8)(half_t)0.f;
  }

  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)(a.M * ROWB), 0x00020000);
  auto issue_tile = [&](int t, int buf) __attribute__((always_inline)) {
    const int ln = lane_id();
    const long left = a.M - (long)t * TP;
    const int npx = left < TP ? (int)left : TP;          // pixels of this tile inside the tensor (the last tile may be partial)
#pragma unroll
    for (int i = 0; i < P_IT; ++i) {
      const int g = wave + NWAVES * i;
      const int c = 64 * g + ln;                         // chunk of the tile: pixel c / 28, slot c % 28 (28 = 0 mod 4: the XOR of
      const int px = __umul24(c, 2341) >> 16;            // the low two bits stays inside the pixel row)
      dma16(rs_x, px < npx ? (c ^ ((px >> 2) & 3)) << 4 : (int)0x80000000, t * TILE_BYTES,
            smem + buf * TILE_BYTES + g * 1024);         // past the tensor: zeros
    }
  };

  float* const stg = (float*)(smem + STG_OFF + q * STG_BLOCK);
  long* const rowt = (long*)(smem + ROWT_OFF) + 32 * q;
  const float rcpW = 1.0f / (float)a.W;
  // stores per wave and tile on the fast path: the pair of a pixel block takes half of the 8 wo 16-byte chunks each
  const int half_chunks = 4 * wo;
  const int nst = (half_chunks + 63) >> 6;
  // fast stores: the 32 rows of a pixel block are one 16-byte aligned run of 32 wo floats (no image boundary inside a block)
  const bool fast = (a.HW & 31) == 0 && ((a.A * wo) & 3) == 0 && ((a.level_off * wo) & 3) == 0;

  int t = blockIdx.x;
  issue_tile(t, 0);
  __builtin_amdgcn_s_waitcnt(0x0070);                  // (the builtin: the compiler does not re-wait for the weight loads in the loop)
  __builtin_amdgcn_s_barrier();

  for (int it = 0;; ++it) {
    const bool more = t + nwg < ntiles;
    if (more) issue_tile(t + nwg, (it + 1) & 1);       // the next tile streams in under this one
    const char* const xin = smem + (it & 1) * TILE_BYTES;
    const long p0 = (long)t * TP + 32 * q;             // first pixel of this wave's block (flat over images)
    const int px = 32 * q + n;                         // pixel of the tile
    const char* const xrow = xin + px * ROWB;
    const int sw = (px >> 2) & 3;
    // anchor of this lane's pixel: image p / HW, anchor index a = p % HW, cell (a / W, a % W)
    const int b0 = (int)(p0 / a.HW);
    int aidx = (int)(p0 - (long)b0 * a.HW) + n, bimg = b0;
    if (aidx >= a.HW) { aidx -= a.HW; ++bimg; }        // (HW >= 32: a block crosses at most one image boundary)
    if (role == 0) {
      float16v acc0 = (float16v)0.f, acc1 = (float16v)0.f;
      half8 bf[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) bf[s] = *(const half8*)(xrow + (((2 * s + h) ^ sw) << 4));
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wv[s], bf[s], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wv[4 + s], bf[s], acc1, 0, 0, 0);
      }
      const float* bias = (const float*)(smem + BIAS_OFF);
      const float d0 = dfl_side(acc0, bias + 16 * h);          // side h (left / top)
      const float d1 = dfl_side(acc1, bias + 32 + 16 * h);     // side 2 + h (right / bottom)
      const int gy = (int)(((float)aidx + 0.5f) * rcpW), gx = aidx - gy * a.W;
      const float ac = (float)(h ? gy : gx) + 0.5f;
      const float c1 = ac - d0, c2 = ac + d1;
      stg[n * wo + h] = (c1 + c2) * 0.5f * a.stride;           // cx / cy
      stg[n * wo + 2 + h] = (c2 - c1) * a.stride;              // w / h
      if (h == 0) rowt[n] = ((long)bimg * a.A + a.level_off + aidx) * wo;
    } else {
      float16v accc = (float16v)0.f, accm = (float16v)0.f;
      half8 bf[4];
#pragma unroll
      for (int g4 = 0; g4 < 2; ++g4) {
#pragma unroll
        for (int s = 0; s < 4; ++s) bf[s] = *(const half8*)(xrow + (((8 + 2 * (4 * g4 + s) + h) ^ sw) << 4));
#pragma unroll
        for (int s = 0; s < 4; ++s) accc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wv[4 * g4 + s], bf[s], accc, 0, 0, 0);
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) bf[s] = *(const half8*)(xrow + (((24 + 2 * s + h) ^ sw) << 4));
#pragma unroll
      for (int s = 0; s < 2; ++s) accm = __builtin_amdgcn_mfma_f32_32x32x16_f16(wv[8 + s], bf[s], accm, 0, 0, 0);
      const float* bias = (const float*)(smem + BIAS_OFF) + 64;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = 16 * h + r;
        if (c < nc) {
          const float z = accc[r] + bias[c];
          stg[n * wo + 4 + c] = 1.0f / (1.0f + __expf(-z));
        }
        if (c < nm) stg[n * wo + 4 + nc + c] = accm[r] + bias[nc + c];
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                      // the rows of every pixel block are complete; the input tile is consumed
    {
      const int ln = lane_id();
      const long nvalid = a.M - p0;                    // rows of this block inside the tensor (>= 32: all)
      if (fast && nvalid >= 32) {
        float* const dst = a.preds + rowt[0];
        for (int i = ln; i < half_chunks; i += 64) {
          const int c = role * half_chunks + i;
          *(float4v*)(dst + 4 * c) = *(const float4v*)(stg + 4 * c);
        }
      } else if (nvalid > 0) {
        const int total = 32 * wo, halfw = total >> 1;
        for (int i = ln; i < halfw; i += 64) {
          const int e = role * halfw + i;
          const int r = e / wo, c = e - r * wo;
          if (r < nvalid) a.preds[rowt[r] + c] = stg[e];
        }
      }
    }
    if (!more) break;
    // the next tile has landed for this wave: its pieces are older than this tile's stores
    if (fast) {
      if (nst == 1) asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory");
      else if (nst == 2) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
      else if (nst == 3) asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");
      else if (nst == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    t += nwg;
  }
}

}  // namespace

// Eligibility: the 224-channel branch tensor (64 box + 128 class + 32 coefficient channels), dense rows, nc <= 32, nm == 32,
// fragment-ordered weights, byte offsets below 2^31.
bool head_tail_ok(const HeadTailArgs& a) {
  if (!a.x || !a.wf || !a.bias || !a.preds || a.ldx != 224 || a.nm != 32 || a.nc < 1 || a.nc > 32) return false;
  if (a.M < 1 || a.HW < 32 || a.W < 1 || a.HW % a.W || a.M % a.HW) return false;
  return a.M * ROWB + TILE_BYTES < (1L << 31);
}

int launch_head_tail(const HeadTailArgs& a, hipStream_t s) {
  if (!head_tail_ok(a)) return -1;
  const int ntiles = (int)((a.M + TP - 1) / TP);
  static int slots = 0;
  if (!slots) {
    hipError_t e = hipFuncSetAttribute((const void*)head_tail_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
      return -2;
    slots = cus > 0 ? cus : 1;
  }
  const int grid = ntiles <= slots ? ntiles : slots;
  hipLaunchKernelGGL(head_tail_kernel, dim3(grid), dim3(64 * NWAVES), LDS_BYTES, s, a, ntiles);
  return (int)hipGetLastError();
}

}  // namespace m355
