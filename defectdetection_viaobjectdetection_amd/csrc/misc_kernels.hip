// HBM-bound kernels of the YOLOv8-seg forward: stem conv (uint8 -> fp16), SPPF pooling, nearest 2x
// upsample into a concat slice, and the Detect decode (DFL softmax-expectation + dist2bbox + sigmoid).
// Replaces upstream aten ops reached from SegmentationModel.forward (SURVEY.md A3/A7/A8/A9;
// call site BscanBased/yolo8_seg_predict.py:8).
#include <stdlib.h>

#include "common.h"

namespace m355 {
namespace {

__device__ __forceinline__ float silu_f(float v) {
  float e = __builtin_amdgcn_exp2f(v * -1.4426950408889634f);
  return v * __builtin_amdgcn_rcpf(1.0f + e);
}

// ---------------------------------------------------------------------------------------------
// Stem: 3x3 stride-2 pad-1 conv on uint8 NHWC (B,H,W,3) -> fp16 NHWC (B,H/2,W/2,COUT), + bias + SiLU.
// HBM-bound layer (1.2 MB in, COUT*0.2 MB out per 640x640 image).  K = 27 (padded to 32) is one
// v_mfma_f32_16x16x32_f16 per 16 channels x 16 pixels: lane (pixel l&15, k-group l>>4) gathers its 8 taps
// as single bytes straight from global memory (the 3x3x3 windows of neighbouring pixels overlap, so the
// bytes come out of L1), converts them to fp16 (0..255 is exact) and feeds them as the B operand; the
// weights [COUT][32] fp16 (unscaled; 1/255 is applied to the fp32 accumulator) are the A operand, loaded
// once per wave.  A wave walks 16-pixel row segments with a grid stride.
// ---------------------------------------------------------------------------------------------
template <int COUT>
__global__ __launch_bounds__(256) void stem_kernel(const StemArgs a) {
  constexpr int MT = COUT / 16;
  const int lane = threadIdx.x & 63;
  const int l15 = lane & 15, g = lane >> 4;
  const int Ho = a.H >> 1, Wo = a.W >> 1;
  const int segs = (Wo + 15) >> 4;                 // 16-pixel segments per output row
  const long total = (long)a.B * Ho * segs;
  const long wave_id = ((long)blockIdx.x * 256 + threadIdx.x) >> 6;
  const long nwaves = ((long)gridDim.x * 256) >> 6;

  half8 wf[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) wf[mt] = *(const half8*)(a.w16 + (mt * 16 + l15) * 32 + g * 8);
  float bias[MT][4];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int j = 0; j < 4; ++j) bias[mt][j] = a.bias[mt * 16 + g * 4 + j];
  // this lane's 8 taps: k = 8g + j -> (kh, q = kw*3 + ci); byte offset relative to (row 2ho-1, col (2wo-1)*3)
  int toff[8];
  int tkh[8];
  bool tok[8];
  int tkw[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = g * 8 + j;
    const int kh = k / 9, q = k - kh * 9;
    tkh[j] = kh;
    tkw[j] = q / 3;
    tok[j] = k < 27;
    toff[j] = kh * a.W * 3 + q;
  }
  const float inv255 = 1.0f / 255.0f;
  for (long t = wave_id; t < total; t += nwaves) {
    const int seg = (int)(t % segs);
    const long r = t / segs;
    const int ho = (int)(r % Ho);
    const int b = (int)(r / Ho);
    const int wo = seg * 16 + l15;
    const bool pv = wo < Wo;
    const int hi0 = 2 * ho - 1, wi0 = 2 * wo - 1;
    const uint8_t* base = a.x + ((long)b * a.H + hi0) * a.W * 3 + (long)wi0 * 3;
    half8 xf;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const bool ok = pv && tok[j] && (unsigned)(hi0 + tkh[j]) < (unsigned)a.H && (unsigned)(wi0 + tkw[j]) < (unsigned)a.W;
      const unsigned v = ok ? (unsigned)base[toff[j]] : 0u;
      xf[j] = (half_t)(float)v;
    }
    float4v acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      acc[mt] = float4v{0.f, 0.f, 0.f, 0.f};
      acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[mt], xf, acc[mt], 0, 0, 0);
    }
    if (pv) {
      half_t* yp = a.y + (long)b * a.y_bstride + ((long)ho * Wo + wo) * a.ldy;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        half4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (half_t)silu_f(acc[mt][j] * inv255 + bias[mt][j]);
        *(half4*)(yp + mt * 16 + g * 4) = o;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Stem, row-staged form (used when a row of W*3 bytes is a multiple of 16): one workgroup per output row.
// The byte gathers of stem_kernel go through the texture path one byte per lane (8 VMEM instructions per
// 16-pixel segment: the kernel ran at 1.7 TB/s).  Here the three input rows of an output row (3 x W x 3 bytes,
// 5.8 KB at W = 640) are staged in LDS with 16-byte coalesced loads -- 16 zero bytes on either side stand for the
// left / right padding, a row outside the image is all zeros -- and the lanes gather their taps with ds_read_u8.
// ---------------------------------------------------------------------------------------------
template <int COUT>
__global__ __launch_bounds__(256) void stem_rows_kernel(const StemArgs a) {
  constexpr int MT = COUT / 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char srow[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  const int Ho = a.H >> 1, Wo = a.W >> 1;
  const int rowb = a.W * 3;          // bytes per input row (multiple of 16)
  const int pitch = rowb + 32;       // LDS row: 16 zero bytes, the row, 16 zero bytes
  const int b = blockIdx.x / Ho, ho = blockIdx.x - b * Ho;
  const int hi0 = 2 * ho - 1;
  {
    const int cpr = pitch / 16;      // 16-byte chunks per LDS row
    for (int c = threadIdx.x; c < 3 * cpr; c += 256) {
      const int r = c / cpr, cc = c - r * cpr;
      const int hi = hi0 + r;
      float4v v = {0.f, 0.f, 0.f, 0.f};
      if (cc > 0 && cc < cpr - 1 && (unsigned)hi < (unsigned)a.H)
        v = *(const float4v*)(a.x + ((long)b * a.H + hi) * rowb + (cc - 1) * 16);
      *(float4v*)(srow + r * pitch + cc * 16) = v;
    }
  }
  half8 wf[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) wf[mt] = *(const half8*)(a.w16 + (mt * 16 + l15) * 32 + g * 8);
  float bias[MT][4];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int j = 0; j < 4; ++j) bias[mt][j] = a.bias[mt * 16 + g * 4 + j];
  // this lane's 8 taps: k = 8g + j -> (kh, q = kw*3 + ci): LDS byte offset relative to the window start
  int toff[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = g * 8 + j;
    const int kh = k / 9, q = k - kh * 9;
    toff[j] = k < 27 ? kh * pitch + q : -1;
  }
  __syncthreads();
  const float inv255 = 1.0f / 255.0f;
  const int segs = (Wo + 15) >> 4;
  for (int seg = wave; seg < segs; seg += 4) {
    const int wo = seg * 16 + l15;
    const bool pv = wo < Wo;
    const int woc = pv ? wo : Wo - 1;
    const unsigned char* base = srow + 16 + (2 * woc - 1) * 3;   // window start (column 2wo-1; -3 bytes = left padding)
    half8 xf;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const unsigned v = toff[j] >= 0 ? (unsigned)base[toff[j]] : 0u;
      xf[j] = (half_t)(float)v;
    }
    float4v acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      acc[mt] = float4v{0.f, 0.f, 0.f, 0.f};
      acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[mt], xf, acc[mt], 0, 0, 0);
    }
    if (pv) {
      half_t* yp = a.y + (long)b * a.y_bstride + ((long)ho * Wo + wo) * a.ldy;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        half4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (half_t)silu_f(acc[mt][j] * inv255 + bias[mt][j]);
        *(half4*)(yp + mt * 16 + g * 4) = o;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// SPPF pooling: y[:, 0:C] = mp5(x), y[:, C:2C] = mp5(mp5(x)), y[:, 2C:3C] = mp5^3(x)  (5x5, s1, p2,
// -inf padding == clipped windows).  One block per (image, 8-channel group); the whole HxW plane of
// 8 channels sits in LDS (6.4 KB at 20x20) and the three pools run back to back from LDS.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ half8 hmax8(half8 a, half8 b) {
  half8 r;
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = a[j] > b[j] ? a[j] : b[j];
  return r;
}

// CG 8-channel groups (CG * 16 bytes of a pixel) per block: consecutive lanes take consecutive groups of one pixel, so
// a wave's loads / stores are runs of CG * 16 contiguous bytes.  Each 5x5 max is separable: a row pass into a scratch
// plane, then a column pass; window indices are clamped (duplicates do not change a max), so both passes are four
// unconditional LDS reads that the compiler unrolls and overlaps.
//
// Generic form (any plane size that fits the LDS): the (row, column) of an element is recomputed in every pass.
template <int CG>
__global__ __launch_bounds__(256) void sppf_pool_generic_kernel(const half_t* x, long x_bstride, int ldx, half_t* y,
                                                                long y_bstride, int ldy, int H, int W, int C) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int n = H * W * CG;
  half8* src = (half8*)smem;
  half8* tmp = src + n;
  half8* dst = tmp + n;
  const int chunks = C / (8 * CG);
  const int b = blockIdx.x / chunks, gq = blockIdx.x % chunks;
  const half_t* xb = x + (long)b * x_bstride + gq * 8 * CG;
  half_t* yb = y + (long)b * y_bstride + gq * 8 * CG;
  for (int i = threadIdx.x; i < n; i += 256) src[i] = *(const half8*)(xb + (long)(i / CG) * ldx + (i % CG) * 8);
  __syncthreads();
  for (int lvl = 0; lvl < 3; ++lvl) {
    for (int i = threadIdx.x; i < n; i += 256) {
      const int px = i / CG, cgi = i - px * CG;
      const int h = px / W, w = px - h * W;
      half8 m = src[i];
#pragma unroll
      for (int d = -2; d <= 2; ++d) {
        const int ww = w + d < 0 ? 0 : (w + d >= W ? W - 1 : w + d);
        m = hmax8(m, src[(h * W + ww) * CG + cgi]);
      }
      tmp[i] = m;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += 256) {
      const int px = i / CG, cgi = i - px * CG;
      const int h = px / W, w = px - h * W;
      half8 m = tmp[i];
#pragma unroll
      for (int d = -2; d <= 2; ++d) {
        const int hh = h + d < 0 ? 0 : (h + d >= H ? H - 1 : h + d);
        m = hmax8(m, tmp[(hh * W + w) * CG + cgi]);
      }
      dst[i] = m;
      *(half8*)(yb + (long)px * ldy + lvl * C + cgi * 8) = m;
    }
    __syncthreads();
    half8* t = src;
    src = dst;
    dst = t;
  }
}

// Planes of at most SPPF_NT * SPPF_ME elements (20 x 20 x 4 groups, 40 x 40 x 1): every thread owns the same <= SPPF_ME elements in
// all six passes, so their (row, column) pairs are divided out once and the passes are clamp + ds_read_b128 + packed fp16 max.
// The generic form spent its time there: two integer divisions and 40 scalar compare/select pairs per element per pass made the
// 26 MB operation VALU-bound (25 us at batch 32; this form: see DESIGN.md section 9).
constexpr int SPPF_ME = 4, SPPF_NT = 512;
__device__ __forceinline__ int clampi(int v, int hi) { return v < 0 ? 0 : (v > hi ? hi : v); }
// (a barrier that orders LDS traffic only: __syncthreads() would also drain the level's global stores before every pass)
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
}

template <int CG>
__global__ __launch_bounds__(SPPF_NT) void sppf_pool_kernel(const half_t* x, long x_bstride, int ldx, half_t* y,
                                                            long y_bstride, int ldy, int H, int W, int C) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int n = H * W * CG, WC = W * CG;
  half8* src = (half8*)smem;
  half8* tmp = src + n;
  half8* dst = tmp + n;
  const int chunks = C / (8 * CG);
  const int b = blockIdx.x / chunks, gq = blockIdx.x % chunks;
  const half_t* xb = x + (long)b * x_bstride + gq * 8 * CG;
  half_t* yb = y + (long)b * y_bstride + gq * 8 * CG;
  // element slots past the plane alias its last element for the reads (no branch around them: the reads of all slots of a
  // pass issue together) and skip the writes
  int ei[SPPF_ME], dr[SPPF_ME][4], dc[SPPF_ME][4], yo[SPPF_ME];
  bool on[SPPF_ME];
#pragma unroll
  for (int k = 0; k < SPPF_ME; ++k) {
    const int i = threadIdx.x + SPPF_NT * k;
    on[k] = i < n;
    ei[k] = on[k] ? i : n - 1;
    const int px = ei[k] / CG, h = px / W, w = px - h * W;
    yo[k] = px * ldy + (ei[k] % CG) * 8;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int d = j < 2 ? j - 2 : j - 1;              // -2, -1, 1, 2
      dr[k][j] = ei[k] + (clampi(w + d, W - 1) - w) * CG;
      dc[k][j] = ei[k] + (clampi(h + d, H - 1) - h) * WC;
    }
    const half8 v = *(const half8*)(xb + (long)px * ldx + (ei[k] % CG) * 8);
    if (on[k]) src[ei[k]] = v;
  }
  lds_barrier();
  for (int lvl = 0; lvl < 3; ++lvl) {
    half8 m[SPPF_ME];
#pragma unroll
    for (int k = 0; k < SPPF_ME; ++k) {
      m[k] = src[ei[k]];
#pragma unroll
      for (int j = 0; j < 4; ++j) m[k] = __builtin_elementwise_max(m[k], src[dr[k][j]]);
    }
#pragma unroll
    for (int k = 0; k < SPPF_ME; ++k)
      if (on[k]) tmp[ei[k]] = m[k];
    lds_barrier();
#pragma unroll
    for (int k = 0; k < SPPF_ME; ++k) {
      m[k] = tmp[ei[k]];
#pragma unroll
      for (int j = 0; j < 4; ++j) m[k] = __builtin_elementwise_max(m[k], tmp[dc[k][j]]);
    }
#pragma unroll
    for (int k = 0; k < SPPF_ME; ++k)
      if (on[k]) {
        dst[ei[k]] = m[k];
        *(half8*)(yb + yo[k] + lvl * C) = m[k];
      }
    lds_barrier();
    half8* t = src;
    src = dst;
    dst = t;
  }
}

// Nearest 2x upsample of an NHWC slice into another NHWC slice (16-byte chunks).
__global__ __launch_bounds__(256) void upsample2x_kernel(const half_t* x, long x_bstride, int ldx, half_t* y,
                                                         long y_bstride, int ldy, int B, int H, int W, int C) {
  const int cg = C / 8;
  const long total = (long)B * 2 * H * 2 * W * cg;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int c = (int)(idx % cg);
    long r = idx / cg;
    const int wo = (int)(r % (2 * W));
    r /= 2 * W;
    const int ho = (int)(r % (2 * H));
    const int b = (int)(r / (2 * H));
    const half8 v = *(const half8*)(x + (long)b * x_bstride + ((long)(ho >> 1) * W + (wo >> 1)) * ldx + c * 8);
    *(half8*)(y + (long)b * y_bstride + ((long)ho * 2 * W + wo) * ldy + c * 8) = v;
  }
}

// ---------------------------------------------------------------------------------------------
// Detect decode (A9).  raw (B,A,64+nc+nm) f32 -> preds (B,A,4+nc+nm) f32.
// One block = 64 consecutive anchors: their raw rows are one contiguous span -> coalesced 16-byte loads
// into LDS; four lanes per anchor compute the DFL expectations (softmax over 16 bins . arange(16)) and
// exchange l,t,r,b by quad shuffles; outputs are staged in LDS and leave as coalesced 16-byte stores.
// ---------------------------------------------------------------------------------------------
constexpr int DEC_ANCHORS = 64;

__global__ __launch_bounds__(256) void head_decode_kernel(const float* raw, long total_anchors, int A, int w3, int w4,
                                                          int w5, int n3, int n4, int nc, int nm, float* preds) {
  extern __shared__ __attribute__((aligned(16))) float dsm[];
  const int wi = 64 + nc + nm, wo = 4 + nc + nm;
  float* sin = dsm;                          // DEC_ANCHORS * wi
  float* sout = dsm + DEC_ANCHORS * wi;      // DEC_ANCHORS * wo
  const long a0 = (long)blockIdx.x * DEC_ANCHORS;
  const int na = (int)((total_anchors - a0 < DEC_ANCHORS) ? (total_anchors - a0) : DEC_ANCHORS);
  const float* src = raw + a0 * wi;
  const int nin = na * wi;
  // a0 is a multiple of 64, so a0*wi*4 bytes is 16-byte aligned for any wi
  for (int i = threadIdx.x * 4; i < nin; i += 256 * 4) {
    if (i + 4 <= nin) {
      *(float4v*)(sin + i) = *(const float4v*)(src + i);
    } else {
      for (int j = i; j < nin; ++j) sin[j] = src[j];
    }
  }
  __syncthreads();
  const int al_blk = threadIdx.x >> 2, q = threadIdx.x & 3;
  const bool valid = al_blk < na;
  const float* rp = sin + (valid ? al_blk : 0) * wi;
  float v[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) v[j] = rp[q * 16 + j];
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
  const float d = sw / se;
  const int a = (int)((a0 + (valid ? al_blk : 0)) % A);
  int al = a, gw = w3;
  float stride = 8.f;
  if (a >= n3 + n4) {
    al = a - n3 - n4; gw = w5; stride = 32.f;
  } else if (a >= n3) {
    al = a - n3; gw = w4; stride = 16.f;
  }
  const int gy = al / gw, gx = al - gy * gw;
  const float ax = (float)gx + 0.5f, ay = (float)gy + 0.5f;
  const int lbase = (threadIdx.x & 63) & ~3;
  const float dl = __shfl(d, lbase + 0), dt = __shfl(d, lbase + 1), dr = __shfl(d, lbase + 2),
              db = __shfl(d, lbase + 3);
  const float x1 = ax - dl, y1 = ay - dt, x2 = ax + dr, y2 = ay + db;
  float o;
  if (q == 0) o = (x1 + x2) * 0.5f;
  else if (q == 1) o = (y1 + y2) * 0.5f;
  else if (q == 2) o = x2 - x1;
  else o = y2 - y1;
  if (valid) {
    float* pp = sout + al_blk * wo;
    pp[q] = o * stride;
    for (int j = q; j < nc; j += 4) {
      const float z = rp[64 + j];
      pp[4 + j] = 1.0f / (1.0f + __expf(-z));
    }
    for (int j = q; j < nm; j += 4) pp[4 + nc + j] = rp[64 + nc + j];
  }
  __syncthreads();
  float* dst = preds + a0 * wo;
  const int nout = na * wo;
  for (int i = threadIdx.x * 4; i < nout; i += 256 * 4) {
    if (i + 4 <= nout) {
      *(float4v*)(dst + i) = *(const float4v*)(sout + i);
    } else {
      for (int j = i; j < nout; ++j) dst[j] = sout[j];
    }
  }
}

}  // namespace

template <int COUT>
static int launch_stem_rows(const StemArgs& a, hipStream_t s) {
  const int lds = 3 * (a.W * 3 + 32);
  hipLaunchKernelGGL(stem_rows_kernel<COUT>, dim3((unsigned)(a.B * (a.H / 2))), dim3(256), lds, s, a);
  return (int)hipGetLastError();
}

int launch_stem(const StemArgs& a, hipStream_t s) {
  if ((a.W * 3) % 16 == 0 && a.W % 2 == 0 && a.H % 2 == 0 && 3 * (a.W * 3 + 32) <= 64 * 1024 && !knobs().stem_gather) {
    switch (a.Cout) {
      case 16: return launch_stem_rows<16>(a, s);
      case 32: return launch_stem_rows<32>(a, s);
      case 48: return launch_stem_rows<48>(a, s);
      case 64: return launch_stem_rows<64>(a, s);
      case 80: return launch_stem_rows<80>(a, s);
      default: return -1;
    }
  }
  const long total = (long)a.B * (a.H / 2) * ((a.W / 2 + 15) / 16);  // one wave per 16-pixel segment
  long blocks = (total + 3) / 4;
  if (blocks > 256 * 32) blocks = 256 * 32;
  const dim3 grid((unsigned)blocks), block(256);
  switch (a.Cout) {
    case 16: hipLaunchKernelGGL(stem_kernel<16>, grid, block, 0, s, a); break;
    case 32: hipLaunchKernelGGL(stem_kernel<32>, grid, block, 0, s, a); break;
    case 48: hipLaunchKernelGGL(stem_kernel<48>, grid, block, 0, s, a); break;
    case 64: hipLaunchKernelGGL(stem_kernel<64>, grid, block, 0, s, a); break;
    case 80: hipLaunchKernelGGL(stem_kernel<80>, grid, block, 0, s, a); break;
    default: return -1;
  }
  return (int)hipGetLastError();
}

int launch_sppf_pool(const half_t* x, long x_bstride, int ldx, half_t* y, long y_bstride, int ldy, int B, int H,
                     int W, int C, hipStream_t s) {
  if (C % 8 || ldx % 8 || ldy % 8) return -1;
  // widest channel chunk per block that still leaves >= 256 blocks (one per CU) and fits the LDS; planes of at most 256 * SPPF_ME
  // elements take the form with per-thread element ownership
  static const int min_blocks = getenv("M355_SPPF_MINBLOCKS") ? atoi(getenv("M355_SPPF_MINBLOCKS")) : 256;
  int cg = 4;
  while (cg > 1 && ((C / 8) % cg != 0 || (long)B * (C / (8 * cg)) < min_blocks || (size_t)3 * H * W * 16 * cg > 160 * 1024 ||
                    H * W * cg > SPPF_NT * SPPF_ME))
    cg >>= 1;
  const size_t lds = (size_t)3 * H * W * 16 * cg;
  if (lds > 160 * 1024) return -1;
  const bool owned = H * W * cg <= SPPF_NT * SPPF_ME && (long)H * W * ldy < (1L << 30);
  auto k = owned ? (cg == 4 ? sppf_pool_kernel<4> : (cg == 2 ? sppf_pool_kernel<2> : sppf_pool_kernel<1>))
                 : (cg == 4 ? sppf_pool_generic_kernel<4> : (cg == 2 ? sppf_pool_generic_kernel<2> : sppf_pool_generic_kernel<1>));
  if (lds > 65536) {
    hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(k, dim3(B * (C / (8 * cg))), dim3(owned ? SPPF_NT : 256), lds, s, x, x_bstride, ldx, y, y_bstride, ldy, H, W, C);
  return (int)hipGetLastError();
}

int launch_upsample2x(const half_t* x, long x_bstride, int ldx, half_t* y, long y_bstride, int ldy, int B, int H,
                      int W, int C, hipStream_t s) {
  if (C % 8 || ldx % 8 || ldy % 8) return -1;
  const long total = (long)B * 4 * H * W * (C / 8);
  long blocks = (total + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(upsample2x_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, x_bstride, ldx, y, y_bstride,
                     ldy, B, H, W, C);
  return (int)hipGetLastError();
}

int launch_head_decode(const float* raw, int B, int in_h, int in_w, int nc, int nm, float* preds, hipStream_t s) {
  const int h3 = in_h / 8, w3 = in_w / 8, h4 = in_h / 16, w4 = in_w / 16, h5 = in_h / 32, w5 = in_w / 32;
  const int n3 = h3 * w3, n4 = h4 * w4, n5 = h5 * w5;
  const int A = n3 + n4 + n5;
  const long total = (long)B * A;
  const size_t lds = (size_t)DEC_ANCHORS * ((64 + nc + nm) + (4 + nc + nm)) * sizeof(float);
  if (lds > 160 * 1024) return -1;
  if (lds > 64 * 1024) {  // many classes (nc = 80: 75 KB)
    hipError_t e = hipFuncSetAttribute((const void*)head_decode_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(head_decode_kernel, dim3((unsigned)((total + DEC_ANCHORS - 1) / DEC_ANCHORS)), dim3(256), lds, s,
                     raw, total, A, w3, w4, w5, n3, n4, nc, nm, preds);
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// ADown's pooling front (yolov9c-seg, SURVEY next row N4; upstream reaches it through F.avg_pool2d / F.max_pool2d):
//   x' = avg_pool2d(x, 2, 1, 0)                          (H-1) x (W-1)
//   a  = x'[..., :C/2]                                   input of the 3x3 / s2 conv
//   m  = max_pool2d(x'[..., C/2:], 3, 2, 1)              (H/2) x (W/2), input of the 1x1 conv (-inf padding)
// HBM-bound and small: one thread per (output pixel, 8-channel group); the averages are taken in fp32 as
// ((p00 + p01) + (p10 + p11)) * 0.25 and rounded to fp16 once (max commutes with that monotone rounding).
// ---------------------------------------------------------------------------------------------------------
namespace {
__device__ __forceinline__ void avg4(const half_t* p, int ldx, int W, float (&o)[8]) {
  const half8 a = *(const half8*)p, b = *(const half8*)(p + ldx), c = *(const half8*)(p + (long)W * ldx),
              d = *(const half8*)(p + (long)W * ldx + ldx);
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (((float)a[j] + (float)b[j]) + ((float)c[j] + (float)d[j])) * 0.25f;
}

__global__ __launch_bounds__(256) void adown_pool_kernel(const half_t* x, long x_bstride, int ldx, half_t* a, long a_bstride, int lda,
                                                         half_t* m, long m_bstride, int ldm, int B, int H, int W, int C) {
  const int cg = C / 16;                                   // 8-channel groups per half
  const long na = (long)B * (H - 1) * (W - 1) * cg, nm = (long)B * (H / 2) * (W / 2) * cg;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < na + nm; i += (long)gridDim.x * 256) {
    if (i < na) {
      const int g = (int)(i % cg);
      long r = i / cg;
      const int xx = (int)(r % (W - 1));
      r /= (W - 1);
      const int yy = (int)(r % (H - 1)), b = (int)(r / (H - 1));
      float v[8];
      avg4(x + b * x_bstride + ((long)yy * W + xx) * ldx + g * 8, ldx, W, v);
      half8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (half_t)v[j];
      *(half8*)(a + b * a_bstride + ((long)yy * (W - 1) + xx) * lda + g * 8) = o;
    } else {
      const long k = i - na;
      const int g = (int)(k % cg);
      long r = k / cg;
      const int Wm = W / 2, Hm = H / 2;
      const int xx = (int)(r % Wm);
      r /= Wm;
      const int yy = (int)(r % Hm), b = (int)(r / Hm);
      float best[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) best[j] = -INFINITY;
      for (int dy = -1; dy <= 1; ++dy) {
        const int ay = 2 * yy + dy;
        if (ay < 0 || ay > H - 2) continue;
        for (int dx = -1; dx <= 1; ++dx) {
          const int ax = 2 * xx + dx;
          if (ax < 0 || ax > W - 2) continue;
          float v[8];
          avg4(x + b * x_bstride + ((long)ay * W + ax) * ldx + C / 2 + g * 8, ldx, W, v);
#pragma unroll
          for (int j = 0; j < 8; ++j) best[j] = fmaxf(best[j], (float)(half_t)v[j]);
        }
      }
      half8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (half_t)best[j];
      *(half8*)(m + b * m_bstride + ((long)yy * Wm + xx) * ldm + g * 8) = o;
    }
  }
}
}  // namespace

int launch_adown_pool(const half_t* x, long x_bstride, int ldx, half_t* a, long a_bstride, int lda, half_t* m, long m_bstride,
                      int ldm, int B, int H, int W, int C, hipStream_t s) {
  if (C % 16 || ldx % 8 || lda % 8 || ldm % 8 || H < 2 || W < 2 || (H & 1) || (W & 1)) return -1;
  const long n = (long)B * ((long)(H - 1) * (W - 1) + (long)(H / 2) * (W / 2)) * (C / 16);
  long blocks = (n + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(adown_pool_kernel, dim3((int)blocks), dim3(256), 0, s, x, x_bstride, ldx, a, a_bstride, lda, m, m_bstride, ldm,
                     B, H, W, C);
  return (int)hipGetLastError();
}

}  // namespace m355
