// model.0 (stem: uint8 -> 32 ch, 3x3 / s2) + model.1 (3x3 / s2, 32 -> 64) + model.2.cv1 (1x1, 64 -> 64) in ONE launch, TWO-TEAM
// form (gfx950; round 3).  Same work, same arithmetic and the same LDS images as conv_stem_s2c32.hip (stage A: the stem on the
// uint8 window -> patch image of the stem output, de-interleaved by column parity; stage B: 3x3 / s2 out of the patch; stage C:
// the 1x1), but the eight waves no longer walk the three stages in lockstep with three barriers per tile (13.4 k cycles per
// tile for ~9.7 k cycles of issue: the LDS round trips of stage B, the barrier skew of 4.5 stem blocks per wave and the
// stage changes were exposed):
//   team X = waves 0-3   stage A of tile i + 1: 36 blocks of 16 patch pixels, nine per wave (gather 8 taps with ds_read_u8,
//                        two 16x16x32 MFMAs, x 1/255 + bias, SiLU, 16 bytes into the patch image of the OTHER buffer); loads
//                        the uint8 window of tile i + 2 meanwhile;
//   team Y = waves 4-7   stages B and C of tile i: wave q owns pixel block q (tile rows 2q, 2q+1) for BOTH 32-channel blocks,
//                        with the 3x3 weights (2 x 18 fragments, 144 VGPRs) in registers for the block's life and the
//                        1x1 weights (2 x 4 fragments) lane-linear in LDS.  The 3x3 weights are in "operand" row order (c2f_c32.hip): lane-half h's
//                        accumulators 8 s .. 8 s + 7 of block b are channels 32 b + 16 s + 8 h .. + 7 = the B fragment of the
//                        1x1's K slice 2 b + s, so the 64-channel intermediate goes from stage B to stage C through fp16
//                        conversion in registers: no Z image, no weights in LDS, no barrier between the stages.
// One barrier per tile: patch(i + 1) is complete and patch(i) is consumed.  Each SIMD holds one X wave (VALU / LDS byte
// gathers / SiLU) and one Y wave (MFMA + SiLU + stores).  Upstream: the first three Conv+BN+SiLU blocks (call
// site BscanBased/yolo8_seg_predict.py:8).  Rounding points and K orders are those of conv_stem_s2c32.hip: bit-identical output.
#include <hip/hip_runtime.h>
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
constexpr int U8_OFF = 2 * PATCH_BYTES;        // two patch images, two windows
constexpr int BIAS_OFF = U8_OFF + 2 * U8_BYTES;    // 64 + 64 + 32 fp32 biases (model.1, model.2.cv1, stem)
constexpr int W0_OFF = BIAS_OFF + 640;         // stem weights [32 rows in MFMA order][32 k] fp16
constexpr int W2_OFF = W0_OFF + 2048;          // the 1x1's 2 x 4 fragments, lane-linear
constexpr int LDS_BYTES = W2_OFF + 8192;       // 115 744
constexpr int NPX = PRR * PCC;                 // 561 patch pixels
static_assert((NPX + 15) / 16 == 36, "nine blocks of 16 patch pixels per wave of team X");

__device__ __forceinline__ void silu16(float16v& v) {       // the five operations of m355_silu per element, staged: same bits
#pragma clang fp contract(off)
  float16v t;
#pragma unroll
  for (int j = 0; j < 16; ++j) t[j] = v[j] * -1.4426950408889634f;
#pragma unroll
  for (int j = 0; j < 16; ++j) t[j] = __builtin_amdgcn_exp2f(t[j]);
#pragma unroll
  for (int j = 0; j < 16; ++j) t[j] = 1.0f + t[j];
#pragma unroll
  for (int j = 0; j < 16; ++j) t[j] = __builtin_amdgcn_rcpf(t[j]);
#pragma unroll
  for (int j = 0; j < 16; ++j) v[j] = v[j] * t[j];
}

__global__ __launch_bounds__(512, 2) void stem_s2c32_cv1_v2_kernel(const ConvArgs a, const StemArgs st, int tiles_x, int tiles_y, int ntiles,
                                                                  int nxb) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int team = wave >> 2, wq = wave & 3;
  const int nwg = gridDim.x;
  const int IH = st.H, IW = st.W, rowb = IW * 3;          // uint8 image

  if (tid < 64) ((float*)(smem + BIAS_OFF))[tid] = a.bias[tid];
  else if (tid < 128) ((float*)(smem + BIAS_OFF))[tid] = a.bias2[tid - 64];
  else if (tid < 160) ((float*)(smem + BIAS_OFF))[tid] = st.bias[tid - 128];
  // stem weights in LDS, row (mt, r) of the two 16-row MFMA tiles = channel (r >> 2) * 8 + mt * 4 + (r & 3): lane group g's
  // accumulators are then channels 8 g .. 8 g + 7
  if (tid >= 256 && tid < 256 + 128) {
    const int i = tid - 256, row = i >> 2, c = i & 3, mt = row >> 4, r = row & 15;
    const int chl = (r >> 2) * 8 + mt * 4 + (r & 3);
    *(float4v*)(smem + W0_OFF + row * 64 + c * 16) = *(const float4v*)(st.w16 + chl * 32 + c * 8);
  }

  auto decode = [&](int vb, int& tb, int& y0, int& x0) __attribute__((always_inline)) {
    const int xcd = vb & 7, q = ntiles >> 3, r = ntiles & 7;
    const int L = (nwg & 7) ? vb : (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (vb >> 3);
    const int tx = L % tiles_x;
    const int rest = L / tiles_x;
    tb = rest / tiles_y;
    y0 = (rest - tb * tiles_y) * TH;
    x0 = tx * TW;
  };

  // ---- uint8 window loader (team X: 256 threads, two of the 35 x 14 chunks each)
  const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)st.x, 0, st.B * IH * rowb, 0x00020000);
  uint4v u8v[2] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};
  auto window_load = [&](int tb, int y0, int x0) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int t = (tid & 255) + 256 * k;
      const int ur = t / UCH, uc = t - ur * UCH;
      uint4v v = {0u, 0u, 0u, 0u};
      const int row = 4 * y0 - 3 + ur;
      const int cb = 12 * x0 - 9 - UDELTA + uc * 16;        // first byte of this chunk inside the image row (may be < 0 / >= rowb)
      if (t < UR * UCH && (unsigned)row < (unsigned)IH) {
        v = __builtin_amdgcn_raw_buffer_load_b128(rs_in, (tb * IH + row) * rowb + cb, 0, 0);
        if (cb < 0 || cb + 16 > rowb) {                       // bytes of the neighbouring row (or past the buffer): the stem's zero padding
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
      u8v[k] = v;
    }
  };
  auto window_store = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int t = (tid & 255) + 256 * k;
      const int ur = t / UCH, uc = t - ur * UCH;
      if (t < UR * UCH) *(uint4v*)(smem + U8_OFF + buf * U8_BYTES + ur * UP + uc * 16) = u8v[k];
    }
  };

  // ---- stage A (team X): 16x16x32 MFMA, lane = (pixel l15 of the block, tap group g)
  const int l15 = lane & 15, g = lane >> 4;
  // blocks blk0, blk0 + 4, ... (nblk of them) of the 36 blocks of 16 patch pixels
  auto stage_a = [&](int y0, int x0, int ubuf_i, int pbuf_i, int blk0, int nblk) __attribute__((always_inline)) {
    int toff[8];   // taps k >= 27 read a valid byte too: their weight rows are zero
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = g * 8 + j;
      const int kh = k / 9, q = k - kh * 9;
      toff[j] = k < 27 ? kh * UP + q + UDELTA : UDELTA;
    }
    const unsigned char* const ubuf = (const unsigned char*)(smem + U8_OFF + ubuf_i * U8_BYTES);
    char* const pimg = smem + pbuf_i * PATCH_BYTES;
    const float inv255 = 1.0f / 255.0f;
    float b0[2][4];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const float4v u = *(const float4v*)(smem + BIAS_OFF + 512 + (8 * g + 4 * mt) * 4);
      b0[mt][0] = u[0]; b0[mt][1] = u[1]; b0[mt][2] = u[2]; b0[mt][3] = u[3];
    }
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
      *(uint4v*)(pimg + dst) = ov;
    };
    const half8 wf0 = *(const half8*)(smem + W0_OFF + l15 * 64 + g * 16);
    const half8 wf1 = *(const half8*)(smem + W0_OFF + (16 + l15) * 64 + g * 16);
    const float4v z4 = {0.f, 0.f, 0.f, 0.f};
    // one block at a time (two interleaved chains, as in the lockstep kernel, cost 40 more registers beside the 144 weight
    // registers every wave of the block is allocated: spills)
    for (int k = 0; k < nblk; ++k) {
      const int blk = blk0 + 4 * k;
      int d0;
      unsigned k0;
      const half8 x0f = gather(blk, d0, k0);
      const float4v a00 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf0, x0f, z4, 0, 0, 0);
      const float4v a01 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf1, x0f, z4, 0, 0, 0);
      finish(a00, a01, d0, k0);
    }
  };
  // Split of the 36 stem blocks of a tile: team X takes NXB per wave, team Y the remaining 9 - NXB per wave in front of its
  // convolutions.  Measured at batch 32 (M355_STEM2_NXB): 9 / 0: 109.8 us, 8 / 1: 108.1, 7 / 2: 107.6 (default), 6 / 3: 112.8;
  // the lockstep kernel: 141.
  const int NXB = nxb;

  // ---- team Y: 3x3 weights in registers (the 1x1's eight fragments in LDS: with them in registers too the compiler spilled 28
  // fragments), fragment offsets of pixel block wq (tile rows 2 wq, 2 wq + 1)
  for (int i = tid; i < 8 * 64; i += 64 * NWAVES) *(float4v*)(smem + W2_OFF + i * 16) = *(const float4v*)(a.wf2 + (long)i * 8);
  half8 w1v[2][18];
  int tb_[3][2];
  const int h = lane >> 5, x15 = lane & 15, r2 = (lane >> 4) & 1;
  if (team == 1) {
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const half_t* wp = a.wf + (long)b * 18 * 512 + lane * 8;
#pragma unroll
      for (int s = 0; s < 18; ++s) w1v[b][s] = *(const half8*)(wp + 512 * s);
    }
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const int j = x15 + (kw >> 1);
      const int row = ((2 * (2 * wq + r2)) * 2 + (kw & 1)) * PJ + j;
#pragma unroll
      for (int s = 0; s < 2; ++s) tb_[kw][s] = row * 64 + (((2 * s + h) ^ ((j >> 2) & 3)) << 4);
    }
  }
  auto stage_bc = [&](int tb, int y0, int x0, int pbuf_i) __attribute__((always_inline)) {
    const char* const pimg = smem + pbuf_i * PATCH_BYTES;
    float16v acc[2];
    {   // bias of model.1 in the operand row order: accumulator r of lane-half h = channel 32 b + 16 (r >> 3) + 8 h + (r & 7)
      const float* bp = (const float*)(smem + BIAS_OFF) + 8 * h;
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const float4v u0 = *(const float4v*)(bp + 32 * b + 16 * s), u1 = *(const float4v*)(bp + 32 * b + 16 * s + 4);
          acc[b][8 * s + 0] = u0[0]; acc[b][8 * s + 1] = u0[1]; acc[b][8 * s + 2] = u0[2]; acc[b][8 * s + 3] = u0[3];
          acc[b][8 * s + 4] = u1[0]; acc[b][8 * s + 5] = u1[1]; acc[b][8 * s + 6] = u1[2]; acc[b][8 * s + 7] = u1[3];
        }
    }
    // stage B: 3x3 / s2, K = 9 taps x 32 channels; one activation fragment feeds both channel blocks
    half8 bf[2][2];
#pragma unroll
    for (int s = 0; s < 2; ++s) bf[0][s] = *(const half8*)(pimg + tb_[0][s]);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      if (tap < 8) {
        const int nk = (tap + 1) / 3, nw = (tap + 1) - 3 * nk;
#pragma unroll
        for (int s = 0; s < 2; ++s) bf[(tap + 1) & 1][s] = *(const half8*)(pimg + tb_[nw][s] + nk * 2 * PJ * 64);
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1v[0][tap * 2 + s], bf[tap & 1][s], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1v[1][tap * 2 + s], bf[tap & 1][s], acc[1], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (a.act) { silu16(acc[0]); silu16(acc[1]); }
    // the fp16 intermediate, already in B-fragment layout: K slice 2 b + s of the 1x1 = accumulators 8 s .. 8 s + 7 of block b
    half8 zf[4];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) zf[2 * b + s][j] = m355_to_half(acc[b][8 * s + j]);
    // stage C: 1x1, K = 64, plain row order (lane-half h's accumulators = channels 32 b + 16 h + r)
    float16v acc2[2];
    {
      const float* bp = (const float*)(smem + BIAS_OFF) + 64 + 16 * h;
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          const float4v u = *(const float4v*)(bp + 32 * b + 4 * qd);
          acc2[b][qd * 4 + 0] = u[0]; acc2[b][qd * 4 + 1] = u[1]; acc2[b][qd * 4 + 2] = u[2]; acc2[b][qd * 4 + 3] = u[3];
        }
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const half8 wa = *(const half8*)(smem + W2_OFF + s * 1024 + lane * 16), wb = *(const half8*)(smem + W2_OFF + (4 + s) * 1024 + lane * 16);
      acc2[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa, zf[s], acc2[0], 0, 0, 0);
      acc2[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wb, zf[s], acc2[1], 0, 0, 0);
    }
    silu16(acc2[0]);
    silu16(acc2[1]);
    const int yy = y0 + 2 * wq + r2, xx = x0 + x15;
    half_t* const yp = (half_t*)a.y + (long)tb * a.y_bstride + ((long)yy * a.Wo + xx) * a.ldy + 16 * h;
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        half8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = m355_to_half(acc2[b][half * 8 + j]);
        *(half8*)(yp + 32 * b + half * 8) = o;
      }
  };

  // ---- prologue: window(0) -> LDS; X: patch(0), window(1) -> LDS
  int vb = blockIdx.x, tb, y0, x0;
  decode(vb, tb, y0, x0);
  if (team == 0) { window_load(tb, y0, x0); window_store(0); }
  __builtin_amdgcn_s_waitcnt(0x0070);              // (the builtin: the compiler does not re-wait for the weight loads in the loop)
  __builtin_amdgcn_s_barrier();
  int nvb = vb + nwg < ntiles ? vb + nwg : -1;
  int ntb = 0, ny0 = 0, nx0 = 0;
  if (nvb >= 0) decode(nvb, ntb, ny0, nx0);
  if (team == 0) {
    if (nvb >= 0) window_load(ntb, ny0, nx0);
    stage_a(y0, x0, 0, 0, wq, NXB);
    if (nvb >= 0) window_store(1);
  } else {
    stage_a(y0, x0, 0, 0, wq + 4 * NXB, 9 - NXB);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                    // patch(0) and window(1) are in LDS

  // iteration it: Y computes tile it out of patch buffer it & 1; X builds patch(it + 1) out of window buffer (it + 1) & 1 into
  // patch buffer (it + 1) & 1 and fetches window(it + 2) into window buffer it & 1 (which it consumed one iteration ago)
  for (int it = 0;; ++it) {
    const bool more = nvb >= 0;
    const int nnvb = more && nvb + nwg < ntiles ? nvb + nwg : -1;
    int nntb = 0, nny0 = 0, nnx0 = 0;
    if (nnvb >= 0) decode(nnvb, nntb, nny0, nnx0);
    if (team == 0) {
      if (more) {
        if (nnvb >= 0) window_load(nntb, nny0, nnx0);
        stage_a(ny0, nx0, (it + 1) & 1, (it + 1) & 1, wq, NXB);
        if (nnvb >= 0) window_store(it & 1);
      }
    } else {
      if (more) stage_a(ny0, nx0, (it + 1) & 1, (it + 1) & 1, wq + 4 * NXB, 9 - NXB);
      stage_bc(tb, y0, x0, it & 1);
    }
    if (!more) break;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    vb = nvb; nvb = nnvb;
    tb = ntb; y0 = ny0; x0 = nx0;
    ntb = nntb; ny0 = nny0; nx0 = nnx0;
  }
}

}  // namespace

// a: the model.1 + model.2.cv1 launch (its x is ignored) with fragment-ordered weights: wf = the 3x3 conv, [channel block][18
// slices] in OPERAND row order; wf2 = the 1x1, [channel block][4 slices] in plain row order.  st: the stem launch.
bool stem_s2c32_v2_ok(const ConvArgs& a, const StemArgs& st) {
  return stem_s2c32_ok(a, st) && a.wf && a.wf2;
}

int launch_stem_s2c32_v2(const ConvArgs& a, const StemArgs& st, hipStream_t s) {
  if (!stem_s2c32_v2_ok(a, st) || !conv_rows_covered(a, 64)) return -1;
  const int tiles_x = a.Wo / TW, tiles_y = a.Ho / TH;
  const int ntiles = st.B * tiles_y * tiles_x;
  static int slots = 0;
  if (!slots) {
    hipError_t e = hipFuncSetAttribute((const void*)stem_s2c32_cv1_v2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
      return -2;
    slots = cus & ~7;
    if (slots < 8) slots = 8;
  }
  const int grid = ntiles <= slots ? ntiles : slots;
  static const int nxb_env = getenv("M355_STEM2_NXB") ? atoi(getenv("M355_STEM2_NXB")) : 7;
  const int nxb = nxb_env < 5 ? 5 : (nxb_env > 9 ? 9 : nxb_env);   // stem blocks per wave of team X (of 9; the rest go to team Y)
  hipLaunchKernelGGL(stem_s2c32_cv1_v2_kernel, dim3(grid), dim3(64 * NWAVES), LDS_BYTES, s, a, st, tiles_x, tiles_y, ntiles, nxb);
  return (int)hipGetLastError();
}

}  // namespace m355
