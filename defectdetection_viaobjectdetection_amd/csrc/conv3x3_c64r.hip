// 3x3 / stride-1 / pad-1 NHWC fp16 convolution with Cin = Cout = 64, WEIGHTS IN REGISTERS (gfx950, v_mfma_f32_32x32x16_f16).
//
// Replaces: the Conv+BN+SiLU 3x3 layers of the 64-channel C2f bottlenecks (YOLOv8s-seg model.4.m.*,
// model.15.m.*) and of the stride-8 box branch (model.22.cv2.0.1) that upstream reaches through
// torch.nn.functional.conv2d (call site: /root/reference/BscanBased/yolo8_seg_predict.py:8).
//
// Why.  On the halo kernel these seven launches ran 27-29 us each for 15.1 GFLOP (0.21 of the MFMA peak): every 8 x 16-pixel
// tile re-streamed the 72 KB weight matrix through LDS-DMA (72 pieces of 1 KiB at 60-180 issue cycles each) and read it
// back fragment by fragment, one ds_read per MFMA on top of the activation fragment.  A 64 x 576 weight matrix is 36
// fragments of a 32-row block: 144 VGPRs.  Here every wave keeps the fragments of ITS channel block in registers for the
// whole life of a persistent block (c2f_c32.hip showed the form): no weight traffic through LDS at all, one ds_read_b128
// per MFMA (the activation fragment), three DMA pieces per wave and tile (the 10 x 18-pixel patch, 23 KB).
//
// Block = 8 waves (two per SIMD), one block per CU, wave (m = wave & 1, q = wave >> 1) = 32 channels x tile rows 2q, 2q+1.
// Per tile: the patch of tile i + 2 is issued, the 36 K slices of tile i run in tap order with the fragments of tap t + 1
// read under the MFMAs of tap t, SiLU (+ residual, read in the accumulator layout at the top of the tile), fp16, and the
// wave's 32 pixels x 32 channels leave through 2 KB of LDS as 64-byte row segments.  One barrier per tile.
// LDS image (as conv3x3_m32.hip): one 128-byte row per patch pixel, pitch 18, the 16-byte chunk index XOR-ed with
// (patch column >> 1) & 7 on the DMA source side and on the reads: conflict free for all nine tap shifts.
#include <stdio.h>
#include <stdlib.h>

#include "common.h"

namespace m355 {
namespace {

typedef float float16v __attribute__((ext_vector_type(16)));

constexpr int TH = 8, TW = 16, PP = 18, ROWB = 128;
constexpr int PROWS = (TH + 2) * PP;                 // 180 patch pixels
constexpr int NPIECES = (PROWS + 7) / 8;             // 23 DMA pieces of 8 rows
constexpr int PATCH_BYTES = NPIECES * 1024;          // 23552
constexpr int NBUF = 3;                              // tile i (compute), i + 1 (landed / landing), i + 2 (being issued)
constexpr int NWAVES = 8;
constexpr int P_IT = (NPIECES + NWAVES - 1) / NWAVES;   // 3
constexpr int STG_OFF = NBUF * PATCH_BYTES;          // output staging: 8 waves x 32 pixels x 64 bytes
constexpr int BIAS_OFF = STG_OFF + NWAVES * 2048;
constexpr int LDS_BYTES = BIAS_OFF + 256;            // 87296

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, int voff, int soff, char* lds) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds, 16, voff, soff, 0, 0);
}

// MFMA row rho = 8 q + 4 h + i is accumulator register 4 q + i of lane-half h; lane-half h's 16 registers = channels 16 h + r
__device__ __forceinline__ int row_plain(int rho) { return 16 * ((rho >> 2) & 1) + 4 * (rho >> 3) + (rho & 3); }

// sixteen SiLUs, staged (the same five operations per element as m355_silu: same bits)
__device__ __forceinline__ void silu16(float16v& v) {
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

__global__ __launch_bounds__(512, 2) void conv3x3_c64r_kernel(const ConvArgs a, int tiles_x, int tiles_y, int ntiles, int sx, int sy,
                                                             int sb) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = a.Hi, W = a.Wi, nwg = gridDim.x;
  const int n = lane & 31, h = lane >> 5;
  const int m = wave & 1, q = wave >> 1;

  if (tid < 64) ((float*)(smem + BIAS_OFF))[tid] = a.bias[tid];

  // ---- this wave's weights: 36 K slices of its 32-channel block, in registers for the block's whole life
  half8 wv[36];
  if (a.wf) {      // fragment-ordered copy: one coalesced 1 KiB load per fragment
    const half_t* wp = a.wf + (long)m * 36 * 512 + lane * 8;
#pragma unroll
    for (int s = 0; s < 36; ++s) wv[s] = *(const half8*)(wp + 512 * s);
  } else {
    const half_t* wp = a.w + (long)(32 * m + row_plain(n)) * a.Kpad + 8 * h;
#pragma unroll
    for (int s = 0; s < 36; ++s) wv[s] = *(const half8*)(wp + 16 * s);
  }

  // ---- tile walk (static, XCD-aware; stepped with carries: no division in the loop -- c2f_c32.hip)
  auto decode = [&](int vb, int& tb, int& ty, int& tx) __attribute__((always_inline)) {
    const int xcd = vb & 7, qq = ntiles >> 3, r = ntiles & 7;
    const int L = (xcd < r ? xcd * (qq + 1) : r * (qq + 1) + (xcd - r) * qq) + (vb >> 3);
    tx = L % tiles_x;
    const int rest = L / tiles_x;
    tb = rest / tiles_y;
    ty = rest - tb * tiles_y;
  };
  auto step_tile = [&](int& tb, int& ty, int& tx) __attribute__((always_inline)) {
    tx += sx;
    if (tx >= tiles_x) { tx -= tiles_x; ++ty; }
    ty += sy;
    if (ty >= tiles_y) { ty -= tiles_y; ++tb; }
    tb += sb;
  };
  const int nimg = a.M / (a.Ho * a.Wo);
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
      (void*)a.x, 0, (int)((nimg - 1) * a.x_bstride + (long)H * W * a.ldx) * 2, 0x00020000);
  const int img_stride = (int)a.x_bstride * 2;

  // ---- patch pieces: wave w owns pieces g = w + 8 i (8 LDS rows each); lane = (row 8 g + lane / 8, chunk slot lane % 8)
  int prel[P_IT], prc[P_IT];
#pragma unroll
  for (int i = 0; i < P_IT; ++i) {
    const int R = 8 * (wave + NWAVES * i) + (lane >> 3);
    const int pr = R / PP, pc = R - pr * PP;
    const int cc = (lane & 7) ^ ((pc >> 1) & 7);
    prel[i] = ((pr * W + pc) * a.ldx + cc * 8) * 2;
    prc[i] = (R < PROWS ? pr : 255) | (pc << 8);              // rows past the patch: never valid
  }
  auto issue_patch = [&](int tb, int y0, int x0, int buf) __attribute__((always_inline)) {
    const int origin = (((y0 - 1) * W + (x0 - 1)) * a.ldx) * 2;   // may be negative for border tiles: those lanes are masked
    const bool interior = y0 >= 1 && y0 + TH + 1 <= H && x0 >= 1 && x0 + TW + 1 <= W;
#pragma unroll
    for (int i = 0; i < P_IT; ++i) {
      const int g = wave + NWAVES * i;
      if (g < NPIECES) {
        const int yy = y0 - 1 + (prc[i] & 255), xx = x0 - 1 + (prc[i] >> 8);
        const bool ok = (prc[i] & 255) != 255 && (interior || ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W));
        dma16(rs_x, ok ? origin + prel[i] : (int)0x80000000, tb * img_stride, smem + buf * PATCH_BYTES + g * 1024);
      }
    }
  };

  // ---- fragment offsets (tile independent): pixel (2 q + (n >> 4), n & 15) of the tile, tap column shift kw, K slice s
  const int r = 2 * q + (n >> 4), c = n & 15;
  int offb[3][4];
#pragma unroll
  for (int kw = 0; kw < 3; ++kw) {
    const int col = c + kw;
#pragma unroll
    for (int s = 0; s < 4; ++s) offb[kw][s] = (r * PP + col) * ROWB + (((2 * s + h) ^ ((col >> 1) & 7)) << 4);
  }
  // output staging (this wave's 32 pixels x 32 channels = 64-byte rows): write chunk (2 h, 2 h + 1) of pixel n, read back
  // pixel 16 i + lane / 4, chunk lane % 4; chunk index XOR (pixel >> 1) & 3
  char* const stg = smem + STG_OFF + wave * 2048;
  const int st_w0 = n * 64 + (((2 * h) ^ ((n >> 1) & 3)) << 4), st_w1 = n * 64 + (((2 * h + 1) ^ ((n >> 1) & 3)) << 4);
  const int st_p = lane >> 2, st_k = lane & 3;

  // Iteration `it`: issue the patch of tile it + 2, compute tile it.  Slot k of the arrays = tile it + k.
  int tbi[3], ty0[3], tx0[3];
  bool have[3];
  int nb_, nty, ntx;
  int vbn = blockIdx.x + nwg;
  decode(blockIdx.x, nb_, nty, ntx);
  have[0] = true;
  tbi[0] = nb_; ty0[0] = nty * TH; tx0[0] = ntx * TW;
  auto plan = [&](int k) __attribute__((always_inline)) {
    have[k] = have[k - 1] && vbn < ntiles;
    if (have[k]) {
      step_tile(nb_, nty, ntx);
      tbi[k] = nb_; ty0[k] = nty * TH; tx0[k] = ntx * TW;
      vbn += nwg;
    }
  };
  plan(1);
  issue_patch(tbi[0], ty0[0], tx0[0], 0);
  if (have[1]) issue_patch(tbi[1], ty0[1], tx0[1], 1);
  // vmcnt(0) lgkmcnt(0) as the BUILTIN: the compiler's wait-count pass sees it and knows the 36 weight loads are done.
  // With an inline-asm wait it re-waited for them inside the tile loop -- vmcnt(35) ... vmcnt(0) in front of the MFMAs of
  // EVERY tile -- and the vmcnt(0) there drained the patch prefetch and the previous tile's stores: 9.7 k cycles per tile
  // instead of 2.5 k (first version of this kernel: 32 us per layer).
  __builtin_amdgcn_s_waitcnt(0x0070);
  __builtin_amdgcn_s_barrier();                            // patches 0 and 1 landed, biases visible

  // diagnostic launches only (a.stamps, M355_STAMPS through m355_conv2d_fwd): cycles per section and wave
  unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = a.stamps ? __builtin_amdgcn_s_memtime() : 0;
  int ntile = 0;
#define C64_STAMP(k)                                                                                      \
  if (a.stamps) {                                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                                    \
    const unsigned long long tn = __builtin_amdgcn_s_memtime();                                           \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                                    \
    tacc[k] += tn - tlast;                                                                                \
    tlast = tn;                                                                                           \
  }
  for (int it = 0;; ++it) {
    plan(2);
    ++ntile;
    const char* const pb = smem + (it % NBUF) * PATCH_BYTES;
    if (have[2]) issue_patch(tbi[2], ty0[2], tx0[2], (it + 2) % NBUF);
    C64_STAMP(0)   // tile step + DMA issue
    // ---- K loop: 9 taps x 4 slices; the fragments of tap t + 1 are read under the MFMAs of tap t
    float16v acc;
    {
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const float4v u = *(const float4v*)(smem + BIAS_OFF + (32 * m + 16 * h + 4 * qd) * 4);
        acc[qd * 4 + 0] = u[0]; acc[qd * 4 + 1] = u[1]; acc[qd * 4 + 2] = u[2]; acc[qd * 4 + 3] = u[3];
      }
    }
    half8 fr[2][4];
#pragma unroll
    for (int s = 0; s < 4; ++s) fr[0][s] = *(const half8*)(pb + offb[0][s]);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      if (tap < 8) {
        const int nk = (tap + 1) / 3, nw = (tap + 1) - 3 * nk;
#pragma unroll
        for (int s = 0; s < 4; ++s) fr[(tap + 1) & 1][s] = *(const half8*)(pb + offb[nw][s] + nk * PP * ROWB);
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wv[4 * tap + s], fr[tap & 1][s], acc, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    C64_STAMP(1)   // reads + MFMAs
    // ---- epilogue: SiLU, + residual, fp16 (the rounding order of the other conv kernels), transpose through LDS, store.
    // The residual (accumulator layout, two 16-byte loads per lane) is loaded HERE, under the SiLUs: the compiler waits for
    // it with vmcnt(0) (the conditional patch issue hides the count from it), which at the top of the tile drained the
    // patch prefetch; by now the pieces issued before the K loop have landed anyway.
    half8 res0 = (half8)(half_t)0.f, res1 = res0;
    if (a.res) {
      const long pix = (long)(ty0[0] + r) * W + tx0[0] + c;
      const half_t* rp = a.res + (long)tbi[0] * a.r_bstride + pix * a.ldr + 32 * m + 16 * h;
      res0 = *(const half8*)rp;
      res1 = *(const half8*)(rp + 8);
    }
    if (a.act) silu16(acc);
    half8 o0, o1;
    {
#pragma clang fp contract(off)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float v0 = acc[j], v1 = acc[8 + j];
        if (a.res) { v0 = v0 + (float)res0[j]; v1 = v1 + (float)res1[j]; }
        o0[j] = m355_to_half(v0);
        o1[j] = m355_to_half(v1);
      }
    }
    C64_STAMP(2)   // residual wait + SiLU + convert
    *(half8*)(stg + st_w0) = o0;
    *(half8*)(stg + st_w1) = o1;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    {
      half_t* const yb = (half_t*)a.y + (long)tbi[0] * a.y_bstride + ((long)(ty0[0] + 2 * q) * W + tx0[0]) * a.ldy + 32 * m;
#pragma unroll
      for (int i = 0; i < 2; ++i) {      // pixels 16 i .. 16 i + 15 of the block = tile row 2 q + i
        const int p = 16 * i + st_p;
        const half8 v = *(const half8*)(stg + p * 64 + ((st_k ^ ((p >> 1) & 3)) << 4));
        *(half8*)(yb + ((long)i * W + st_p) * a.ldy + st_k * 8) = v;
      }
    }
    // the patch of tile it + 1 (issued one iteration ago, or in the prologue) has landed for this wave: everything older than
    // this iteration's own patch pieces and stores (the residual loads were consumed above)
    C64_STAMP(3)   // staging + stores
    if (have[2]) {
      if (wave + 2 * NWAVES < NPIECES) asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)" ::: "memory");   // 3 pieces + 2 stores
      else asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");                                // 2 pieces + 2 stores
    } else {
      asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
    }
    C64_STAMP(4)   // next patch landed
    __builtin_amdgcn_s_barrier();
    C64_STAMP(5)   // barrier
    if (!have[1]) break;
#pragma unroll
    for (int k = 0; k < 2; ++k) { tbi[k] = tbi[k + 1]; ty0[k] = ty0[k + 1]; tx0[k] = tx0[k + 1]; have[k] = have[k + 1]; }
  }
  if (a.stamps && lane == 0) {
    unsigned long long* o = a.stamps + ((long)blockIdx.x * NWAVES + wave) * 8;
    for (int k = 0; k < 6; ++k) o[k] = tacc[k];
    o[6] = (unsigned long long)ntile;
  }
#undef C64_STAMP
}

}  // namespace

// Eligibility: 3x3 / s1 / p1, Cin = Cout = 64, fp16 out, map a multiple of the 8 x 16 tile, 31-bit byte offsets.
bool conv3x3_c64r_ok(const ConvArgs& a) {
  if (a.ksize != 3 || a.stride != 1 || a.pad != 1 || a.out_f32 || a.convt_co > 0 || a.tmode || a.phase || a.csplit || a.w2 || a.dec_preds)
    return false;
  if (a.Cin != 64 || a.Cout != 64 || a.ldx % 8 || a.ldy % 8 || a.Kpad < 576 || a.Kpad % 8) return false;
  if (a.Ho != a.Hi || a.Wo != a.Wi || a.Hi % TH || a.Wi % TW) return false;
  if (a.res && a.ldr % 8) return false;
  const long nimg = a.Ho * a.Wo > 0 ? a.M / ((long)a.Ho * a.Wo) : 0;
  if (nimg < 1) return false;
  return ((nimg - 1) * a.x_bstride + (long)a.Hi * a.Wi * a.ldx) * 2 < (1L << 31);
}

int launch_conv3x3_c64r(const ConvArgs& a, hipStream_t s) {
  if (!conv3x3_c64r_ok(a) || !conv_rows_covered(a, 64)) return -1;
  const int tiles_x = a.Wi / TW, tiles_y = a.Hi / TH;
  const int B = a.M / (a.Ho * a.Wo);
  const int ntiles = B * tiles_y * tiles_x;
  static int slots = 0;
  if (!slots) {
    hipError_t e = hipFuncSetAttribute((const void*)conv3x3_c64r_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
      return -2;
    slots = cus & ~7;   // one block per CU; the XCD-aware tile order needs gridDim.x % 8 == 0 whenever a block walks > 1 tile
    if (slots < 8) slots = 8;
  }
  const int grid = ntiles <= slots ? ntiles : slots;
  const int step = grid >> 3;
  const int sx = step % tiles_x, sy = (step / tiles_x) % tiles_y, sb = step / tiles_x / tiles_y;
  hipLaunchKernelGGL(conv3x3_c64r_kernel, dim3(grid), dim3(64 * NWAVES), LDS_BYTES, s, a, tiles_x, tiles_y, ntiles, sx, sy, sb);
  return (int)hipGetLastError();
}

}  // namespace m355
