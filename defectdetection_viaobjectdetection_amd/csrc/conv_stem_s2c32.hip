// model.0 (stem: uint8 -> 32 ch, 3x3 / s2) + model.1 (3x3 / s2, 32 -> 64) + model.2.cv1 (1x1, 64 -> 64) in ONE launch (gfx950).
// Replaces the first three Conv+BN+SiLU blocks upstream reaches through F.conv2d (call site
// BscanBased/yolo8_seg_predict.py:8).
//
// The stem's output is the largest tensor of the network (320 x 320 x 32 fp16 = 6.6 MB per image: 210 MB at batch 32) and has
// one consumer.  As two launches it is written (stem, 66 us) and read back (conv3x3_s2c32.hip, 90 us): 420 MB of the forward's
// 4.9 GB of HBM traffic.  Here the patch kernel of conv3x3_s2c32.hip computes its own 17 x 33-pixel patch of the stem output
// from a 35 x 69-pixel window of the uint8 image (7 KB instead of 36 KB per tile; 10 % of the stem is computed twice at the
// tile seams) and the stem output never exists in memory.
//
// Per 8 x 16 output tile, 8 waves:
//   stage A  stem: the uint8 window of this tile sits in LDS (16-byte loads of the NEXT tile's window are issued at the top
//            of a tile into registers and written to the other of two 8 KB buffers at its end).  36 blocks of 16 patch pixels:
//            lanes gather their 8 taps with ds_read_u8 (k = (kh * 3 + kw) * 3 + c, as stem_rows_kernel), one
//            v_mfma_f32_16x16x32_f16 per 16 channels, x 1/255 + bias, SiLU, fp16, 16 bytes (8 channels) per lane into the
//            patch image -- rows de-interleaved by column parity, chunks swizzled, exactly the image conv3x3_s2c32.hip's
//            LDS-DMA produced, zeros where the patch leaves the 320 x 320 map (the padding of model.1).
//   stage B  3x3 / s2 out of the patch, stage C 1x1 through LDS: as conv3x3_s2c32.hip (32x32x16 MFMA, 32 ch x 32 px per wave).
// Three barriers per tile (patch complete, intermediate complete, next window written / everyone done with this tile).
#include <stdio.h>
#include <stdlib.h>

#include "common.h"

namespace m355 {
namespace {

typedef float float16v __attribute__((ext_vector_type(16)));
typedef unsigned uint4v __attribute__((ext_vector_type(4)));

constexpr int TH = 8, TW = 16;                 // output tile (160 x 160 map at 640 x 640)
constexpr int PRR = 2 * TH + 1, PCC = 2 * TW + 1;   // 17 x 33 patch of the stem output
constexpr int PJ = 20;                         // pixel pitch of a (row, parity) plane of the patch image
constexpr int PATCH_BYTES = 43 * 1024;         // 680 rows of 64 bytes, as conv3x3_s2c32.hip
constexpr int UR = 4 * TH + 3;                 // 35 rows of the uint8 window
constexpr int UCH = 14;                        // 16-byte chunks per window row: 7 + 69 * 3 = 214 <= 224 bytes
constexpr int UP = 240;                        // LDS pitch of a window row
constexpr int UDELTA = 7;                      // the window's first byte inside its first chunk: (12 x0 - 9) mod 16, x0 % 16 == 0
constexpr int U8_BYTES = UR * UP;              // 8400
constexpr int NWAVES = 8;
constexpr int W1_PITCH = 592, W2_PITCH = 144, Z_PITCH = 144;
constexpr int U8_OFF = PATCH_BYTES, W1_OFF = U8_OFF + 2 * U8_BYTES, W2_OFF = W1_OFF + 64 * W1_PITCH, Z_OFF = W2_OFF + 64 * W2_PITCH;
constexpr int BIAS_OFF = Z_OFF + 128 * Z_PITCH;    // 64 + 64 + 32 fp32 biases (model.1, model.2.cv1, stem)
constexpr int W0_OFF = BIAS_OFF + 640;          // stem weights [32 rows in MFMA order][32 k] fp16
constexpr int NSLOT_OFF = W0_OFF + 2048;
constexpr int LDS_BYTES = NSLOT_OFF + 16;
constexpr int NPX = PRR * PCC;                 // 561 patch pixels
constexpr int NBLK = (NPX + 15) / 16;          // 36 blocks of 16

__device__ __forceinline__ int chl_of(int R) {   // as conv3x3_s2c32.hip: lane-half h's 16 accumulators = 16 consecutive channels
  const int rho = R & 31;
  return (R & ~31) + 16 * ((rho >> 2) & 1) + 4 * (rho >> 3) + (rho & 3);
}

__global__ __launch_bounds__(512, 2) void stem_s2c32_cv1_kernel(const ConvArgs a, const StemArgs st, int tiles_x, int tiles_y, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nwg = gridDim.x;
  const int IH = st.H, IW = st.W, rowb = IW * 3;          // uint8 image

  for (int i = tid; i < 64 * 36; i += 64 * NWAVES) {
    const int R = i / 36, c = i - R * 36;
    *(float4v*)(smem + W1_OFF + R * W1_PITCH + c * 16) = *(const float4v*)(a.w + (long)chl_of(R) * a.Kpad + c * 8);
  }
  for (int i = tid; i < 64 * 8; i += 64 * NWAVES) {
    const int R = i >> 3, c = i & 7;
    *(float4v*)(smem + W2_OFF + R * W2_PITCH + c * 16) = *(const float4v*)(a.w2 + (long)chl_of(R) * 64 + c * 8);
  }
  if (tid < 64) ((float*)(smem + BIAS_OFF))[tid] = a.bias[tid];
  else if (tid < 128) ((float*)(smem + BIAS_OFF))[tid] = a.bias2[tid - 64];
  else if (tid < 160) ((float*)(smem + BIAS_OFF))[tid] = st.bias[tid - 128];
  // stem weights in LDS, row (mt, r) of the two 16-row MFMA tiles = channel (r >> 2) * 8 + mt * 4 + (r & 3): lane group g's
  // accumulators are then channels 8 g .. 8 g + 7 (hipcc re-loads loop-invariant global values inside the tile loop: the
  // register copy of these 2 KB cost a vmcnt(0) -- the next window's load and the previous tile's stores -- per block)
  if (tid >= 256 && tid < 256 + 128) {
    const int i = tid - 256, row = i >> 2, c = i & 3, mt = row >> 4, r = row & 15;
    const int chl = (r >> 2) * 8 + mt * 4 + (r & 3);
    *(float4v*)(smem + W0_OFF + row * 64 + c * 16) = *(const float4v*)(st.w16 + chl * 32 + c * 8);
  }

  int* const tq = a.tileq;
  auto decode = [&](int vb, int& tb, int& y0, int& x0) __attribute__((always_inline)) {
    const int xcd = vb & 7, q = ntiles >> 3, r = ntiles & 7;
    const int L = tq ? vb : (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (vb >> 3);
    const int tx = L % tiles_x;
    const int rest = L / tiles_x;
    tb = rest / tiles_y;
    y0 = (rest - tb * tiles_y) * TH;
    x0 = tx * TW;
  };

  // ---- uint8 window loader: thread t < 35 * 14 owns chunk (row t / 14, chunk t % 14) of every tile's window
  const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)st.x, 0, st.B * IH * rowb, 0x00020000);
  const int ur = tid / UCH, uc = tid - ur * UCH;
  const bool uload = tid < UR * UCH;
  uint4v u8v = {0u, 0u, 0u, 0u};
  auto window_load = [&](int tb, int y0, int x0) __attribute__((always_inline)) {
    if (!uload) return;
    const int row = 4 * y0 - 3 + ur;
    const int cb = 12 * x0 - 9 - UDELTA + uc * 16;          // first byte of this chunk inside the image row (may be < 0 / >= rowb)
    uint4v v = {0u, 0u, 0u, 0u};
    if ((unsigned)row < (unsigned)IH) {
      v = __builtin_amdgcn_raw_buffer_load_b128(rs_in, (tb * IH + row) * rowb + cb, 0, 0);
      if (cb < 0 || cb + 16 > rowb) {                         // bytes of the neighbouring row (or past the buffer): the stem's zero padding
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          unsigned m = 0;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int bi = cb + 4 * w + j;
            if (bi >= 0 && bi < rowb) m |= 0xffu << (8 * j);
          }
          v[w] &= m;
        }
      }
    }
    u8v = v;
  };
  auto window_store = [&](int buf) __attribute__((always_inline)) {
    if (uload) *(uint4v*)(smem + U8_OFF + buf * U8_BYTES + ur * UP + uc * 16) = u8v;
  };

  // ---- stage A constants: 16x16x32 MFMA, lane = (pixel l15 of the block, tap group g)
  const int l15 = lane & 15, g = lane >> 4;
  int toff[8];   // taps k >= 27 read a valid byte too: their weight rows are zero
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = g * 8 + j;
    const int kh = k / 9, q = k - kh * 9;
    toff[j] = k < 27 ? kh * UP + q + UDELTA : UDELTA;
  }

  // ---- stage B / C fragment offsets (conv3x3_s2c32.hip)
  const int l31 = lane & 31, h = lane >> 5, x15 = lane & 15, r2 = (lane >> 4) & 1;
  const int wq = wave & 3, wm = wave >> 2;
  int tb_[3][2];
#pragma unroll
  for (int kw = 0; kw < 3; ++kw) {
    const int j = x15 + (kw >> 1);
    const int row = ((2 * (2 * wq + r2)) * 2 + (kw & 1)) * PJ + j;
#pragma unroll
    for (int s = 0; s < 2; ++s) tb_[kw][s] = row * 64 + (((2 * s + h) ^ ((j >> 2) & 3)) << 4);
  }
  const int ta1 = W1_OFF + (wm * 32 + l31) * W1_PITCH + h * 16;
  const int ta2 = W2_OFF + (wm * 32 + l31) * W2_PITCH + h * 16;
  const int tz = Z_OFF + (wq * 32 + l31) * Z_PITCH;
  const int tbias = BIAS_OFF + 16 * h * 4;
  auto bias_vec = [&](int which) __attribute__((always_inline)) {
    float16v v;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4v u = *(const float4v*)(smem + tbias + which * 256 + (wm * 32 + q * 4) * 4);
      v[q * 4 + 0] = u[0]; v[q * 4 + 1] = u[1]; v[q * 4 + 2] = u[2]; v[q * 4 + 3] = u[3];
    }
    return v;
  };

  int* const nslot = (int*)(smem + NSLOT_OFF);
  int claim = 0;
  auto claim_issue = [&]() __attribute__((always_inline)) {
    if (wave == 0 && lane == 0) claim = __hip_atomic_fetch_add(tq, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  auto claim_publish = [&]() __attribute__((always_inline)) {
    if (wave == 0 && lane == 0) {
      const int L = nwg + claim;
      *nslot = L < ntiles ? L : -1;
    }
  };

  int vb = blockIdx.x, tb, y0, x0;
  if (tq) claim_issue();
  decode(vb, tb, y0, x0);
  window_load(tb, y0, x0);
  window_store(0);
  if (tq) claim_publish();
  __syncthreads();
  int nvb = tq ? __builtin_amdgcn_readfirstlane(*nslot) : (vb + nwg < ntiles ? vb + nwg : -1);

  const float inv255 = 1.0f / 255.0f;
  for (int it = 0;; ++it) {
    const int ub = it & 1;
    const bool more = nvb >= 0;
    int ntb = 0, ny0 = 0, nx0 = 0;
    if (more) {                                   // the next tile's window: one 16-byte load per thread, in flight under this tile
      decode(nvb, ntb, ny0, nx0);
      window_load(ntb, ny0, nx0);
      if (tq) claim_issue();
    }
    // ---- stage A: the stem on this tile's window -> patch image
    {
      const unsigned char* const ubuf = (const unsigned char*)(smem + U8_OFF + ub * U8_BYTES);
      float b0[2][4];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const float4v u = *(const float4v*)(smem + BIAS_OFF + 512 + (8 * g + 4 * mt) * 4);
        b0[mt][0] = u[0]; b0[mt][1] = u[1]; b0[mt][2] = u[2]; b0[mt][3] = u[3];
      }
      // one block of 16 patch pixels: gather (8 taps per lane), two MFMAs, SiLU, one 16-byte write into the patch image
      auto gather = [&](int blk, int& dst, unsigned& keep) __attribute__((always_inline)) -> half8 {
        const int p = blk * 16 + l15;
        const bool pv = p < NPX;
        const int pp = pv ? p : NPX - 1;
        const int pr = pp / PCC, pc = pp - pr * PCC;
        const unsigned char* base = ubuf + (2 * pr) * UP + (2 * pc) * 3;
        half8 xf;
#pragma unroll
        for (int j = 0; j < 8; ++j) xf[j] = (half_t)(float)(unsigned)base[toff[j]];
        // patch pixel (pr, pc) = stem pixel (2 y0 - 1 + pr, 2 x0 - 1 + pc): outside the map only on the first row / column ->
        // zeros (the padding of model.1), branch-free: the 16 bytes are ANDed with a mask
        keep = ((y0 == 0 && pr == 0) || (x0 == 0 && pc == 0)) ? 0u : 0xffffffffu;
        const int jj = pc >> 1;
        const int R = pv ? (pr * 2 + (pc & 1)) * PJ + jj : PJ - 1;       // pixels past the patch: a pad row nobody reads (plane 0, j = 19)
        dst = R * 64 + ((g ^ ((jj >> 2) & 3)) << 4);
        return xf;
      };
      auto finish = [&](const float4v& acc0, const float4v& acc1, int dst, unsigned keep) __attribute__((always_inline)) {
        half8 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          o[j] = m355_to_half(m355_silu(acc0[j] * inv255 + b0[0][j]));
          o[4 + j] = m355_to_half(m355_silu(acc1[j] * inv255 + b0[1][j]));
        }
        uint4v ov = *(const uint4v*)&o;
        ov[0] &= keep; ov[1] &= keep; ov[2] &= keep; ov[3] &= keep;
        *(uint4v*)(smem + dst) = ov;
      };
      const half8 wf0 = *(const half8*)(smem + W0_OFF + l15 * 64 + g * 16);
      const half8 wf1 = *(const half8*)(smem + W0_OFF + (16 + l15) * 64 + g * 16);
      const float4v z4 = {0.f, 0.f, 0.f, 0.f};
      for (int blk = wave; blk < NBLK; blk += 2 * NWAVES) {     // two blocks per trip: their read -> convert -> MFMA -> SiLU chains interleave
        const bool two = blk + NWAVES < NBLK;                    // (uniform per wave)
        int d0, d1;
        unsigned k0, k1;
        const half8 x0f = gather(blk, d0, k0);
        const half8 x1f = gather(two ? blk + NWAVES : blk, d1, k1);
        const float4v a00 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf0, x0f, z4, 0, 0, 0);
        const float4v a01 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf1, x0f, z4, 0, 0, 0);
        const float4v a10 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf0, x1f, z4, 0, 0, 0);
        const float4v a11 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf1, x1f, z4, 0, 0, 0);
        finish(a00, a01, d0, k0);
        if (two) finish(a10, a11, d1, k1);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                   // the patch is complete (raw barrier: the window load and the stores stay in flight)
    // ---- stage B: 3x3 / s2, K = 9 taps x 32 channels
    float16v acc = bias_vec(0);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int kh = tap / 3, kw = tap - 3 * kh;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const half8 bf = *(const half8*)(smem + tb_[kw][s] + kh * 2 * PJ * 64);
        const half8 af = *(const half8*)(smem + ta1 + tap * 64 + s * 32);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bf, acc, 0, 0, 0);
      }
    }
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      half8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float v = acc[half * 8 + j];
        if (a.act) v = m355_silu(v);
        o[j] = m355_to_half(v);
      }
      *(half8*)(smem + tz + (wm * 32 + 16 * h + half * 8) * 2) = o;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                   // both channel halves of every intermediate row are in LDS
    // ---- stage C: 1x1, K = 64
    float16v acc2 = bias_vec(1);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const half8 bf = *(const half8*)(smem + tz + s * 32 + h * 16);
      const half8 af = *(const half8*)(smem + ta2 + s * 32);
      acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bf, acc2, 0, 0, 0);
    }
    int nnvb = -1;
    if (more) {
      window_store(ub ^ 1);                         // (waits for this thread's window load only)
      if (tq) claim_publish();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                 // the next window is in LDS; every wave is done with the patch and the intermediate
      nnvb = tq ? __builtin_amdgcn_readfirstlane(*nslot) : (nvb + nwg < ntiles ? nvb + nwg : -1);
    }
    {
      const int yy = y0 + 2 * wq + r2, xx = x0 + x15;
      half_t* const yp = (half_t*)a.y + (long)tb * a.y_bstride + ((long)yy * a.Wo + xx) * a.ldy + wm * 32 + 16 * h;
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        half8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = m355_to_half(m355_silu(acc2[half * 8 + j]));
        *(half8*)(yp + half * 8) = o;
      }
    }
    if (!more) break;
    vb = nvb; nvb = nnvb;
    tb = ntb; y0 = ny0; x0 = nx0;
  }
  if (tq && tid == 0) {
    if (__hip_atomic_fetch_add(tq + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nwg - 1) {
      __hip_atomic_store(tq, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(tq + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

}  // namespace

// a: the model.1 + model.2.cv1 launch of conv3x3_s2c32.hip (its x is ignored); st: the stem launch (its y is ignored).
bool stem_s2c32_ok(const ConvArgs& a, const StemArgs& st) {
  static const bool off = getenv("M355_NO_STEMFUSE") != nullptr;
  if (off || !conv_s2c32_cv1_ok(a)) return false;
  if (st.Cout != 32 || st.H != 2 * a.Hi || st.W != 2 * a.Wi || (st.W * 3) % 16 || !st.w16 || !st.bias) return false;
  return (long)st.B * st.H * st.W * 3 < (1L << 31) && st.B == a.M / (a.Ho * a.Wo);
}

int launch_stem_s2c32(const ConvArgs& a, const StemArgs& st, hipStream_t s) {
  if (!stem_s2c32_ok(a, st) || !conv_rows_covered(a, 64)) return -1;
  const int tiles_x = a.Wo / TW, tiles_y = a.Ho / TH;
  const int ntiles = st.B * tiles_y * tiles_x;
  static int slots = 0;
  if (!slots) {
    hipError_t e = hipFuncSetAttribute((const void*)stem_s2c32_cv1_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
      return -2;
    slots = cus & ~7;
    if (slots < 8) slots = 8;
  }
  const int grid = ntiles <= slots ? ntiles : slots;
  hipLaunchKernelGGL(stem_s2c32_cv1_kernel, dim3(grid), dim3(64 * NWAVES), LDS_BYTES, s, a, st, tiles_x, tiles_y, ntiles);
  return (int)hipGetLastError();
}

}  // namespace m355
