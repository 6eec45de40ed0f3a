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
// whole life of a persistent block: no weight traffic through LDS at all, one ds_read_b128 per MFMA (the activation
// fragment), the 10 x 18-pixel patch (23 KB) by LDS-DMA.
//
// Block = 8 waves (two per SIMD), one block per CU, wave (m = wave & 1, q = wave >> 1) = 32 channels x tile rows 2q, 2q+1.
// First form (round 3, 27.5 us -- no better than the halo kernel): all eight waves in lockstep, one barrier per tile; the stamps
// showed 5.1 k cycles per tile for 2.3 k cycles of MFMA: the two waves of a SIMD were in the K loop together (half the pipe
// each), then in the SiLUs together, then in the stores.  This form (the schedule of conv3x3_s2c64.hip): TEAM t = wave >> 2
// owns tile rows 4t .. 4t+3 and the teams run half a tile apart,
//   slot 1   team 0: K(i)                           team 1: epilogue(i - 1), + part of the patch of tile i + 1
//   slot 2   team 0: epilogue(i), + rest of patch   team 1: K(i)
// one barrier after each slot, so the MFMA pipe of a SIMD serves one wave at a time and the other wave's SiLU / stores run
// beside it.  Epilogue: SiLU (+ residual, loaded in the accumulator layout BEFORE the DMA issue so that the wait for it does
// not wait for the pieces behind it), fp16, 2 KB of LDS per wave, 64-byte row segments out.  The DMA is issued
// unconditionally (a tile that does not exist is all out-of-range offsets: zeros into the idle buffer) so that every wait
// count is static.
// LDS image (as conv3x3_m32.hip): one 128-byte row per patch pixel, pitch 18, the 16-byte chunk index XOR-ed with
// (patch column >> 1) & 7 on the DMA source side and on the reads: conflict free for all nine tap shifts.
#include <hip/hip_runtime.h>
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
constexpr int NWAVES = 8;
constexpr int G_SPLIT = 11, P_IT = 3;                // pieces 0-10: team 1 (slot 1), 11-22: team 0 (slot 2); per wave
constexpr int STG_OFF = 2 * PATCH_BYTES;             // output staging: 8 waves x 32 pixels x 64 bytes
constexpr int BIAS_OFF = STG_OFF + NWAVES * 2048;
constexpr int LDS_BYTES = BIAS_OFF + 256;            // 63744

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

__device__ __forceinline__ int lane_id() {            // volatile: lane-derived values are rebuilt where they are used, not kept
  int ln;                                             // live (= spilled) across the K loop; a scratch reload waits on vmcnt(0)
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
  return ln;
}

__global__ __launch_bounds__(512, 2) void conv3x3_c64r_kernel(const ConvArgs a, int tiles_x, int tiles_y, int ntiles, int sx, int sy,
                                                             int sb) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = a.Hi, W = a.Wi, nwg = gridDim.x;
  const int n = lane & 31, h = lane >> 5;
  const int m = wave & 1, q = wave >> 1, team = wave >> 2;

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

  // ---- tile walk (static, XCD-aware; decoded once, then stepped with carries)
  auto decode = [&](int vb, int& tb, int& ty, int& tx) __attribute__((always_inline)) {
    const int xcd = vb & 7, qq = ntiles >> 3, r = ntiles & 7;
    const int L = (nwg & 7) ? vb : (xcd < r ? xcd * (qq + 1) : r * (qq + 1) + (xcd - r) * qq) + (vb >> 3);
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

  // ---- patch pieces (8 LDS rows each): wave w of a team owns pieces g = g0 + (w & 3) + 4 i; lane = (row 8 g + lane / 8, chunk
  // slot lane % 8).  `valid` false: every lane out of range (zeros into the idle buffer), same instruction count.
  auto issue_patch = [&](bool valid, int tb, int y0, int x0, int buf, int g0, int g1) __attribute__((always_inline)) {
    const int ln = lane_id();
    const int r0 = 8 * (g0 + (wave & 3)) + (ln >> 3), slot = ln & 7;
    const int origin = (__mul24(y0 - 1, W) + x0 - 1) * a.ldx * 2;     // may be negative for border tiles: those lanes are masked
#pragma unroll
    for (int i = 0; i < P_IT; ++i) {
      const int g = g0 + (wave & 3) + 4 * i;
      if (g < g1) {
        const int R = r0 + 32 * i;
        const int pr = __umul24(R, 3641) >> 16, pc = R - pr * PP;     // R / 18, R % 18 (exact for R < 200)
        const int yy = y0 - 1 + pr, xx = x0 - 1 + pc;
        const bool ok = valid && R < PROWS && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
        const int rel = (__mul24(__mul24(pr, W) + pc, a.ldx) + ((slot ^ ((pc >> 1) & 7)) << 3)) * 2;
        dma16(rs_x, ok ? origin + rel : (int)0x80000000, tb * img_stride, smem + buf * PATCH_BYTES + g * 1024);
      }
    }
  };

  // ---- fragment offsets (tile independent): pixel (2 q + (n >> 4), n & 15) of the tile, tap column shift kw, K slice s
  int offb[3][4];
  {
    const int r = 2 * q + (n >> 4), c = n & 15;
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const int col = c + kw;
#pragma unroll
      for (int s = 0; s < 4; ++s) offb[kw][s] = (r * PP + col) * ROWB + (((2 * s + h) ^ ((col >> 1) & 7)) << 4);
    }
  }
  char* const stg = smem + STG_OFF + wave * 2048;

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

  float16v acc;
  // K: 9 taps x 4 slices out of patch buffer pb; the fragments of tap t + 1 are read under the MFMAs of tap t
  auto k_loop = [&](const char* pb) __attribute__((always_inline)) {
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
      const float4v u = *(const float4v*)(smem + BIAS_OFF + (32 * m + 16 * h + 4 * qd) * 4);
      acc[qd * 4 + 0] = u[0]; acc[qd * 4 + 1] = u[1]; acc[qd * 4 + 2] = u[2]; acc[qd * 4 + 3] = u[3];
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
  };
  // epilogue of tile (tb, y0, x0) + the issue of this team's part of the next patch between the residual loads and their use
  auto epilogue = [&](int tb, int y0, int x0, bool nvalid, int ntb, int ny0, int nx0, int nbuf, int g0, int g1) __attribute__((always_inline)) {
    const int ln = lane_id();
    const int en = ln & 31, eh = ln >> 5, st_p = ln >> 2, st_k = ln & 3;
    half8 res0 = (half8)(half_t)0.f, res1 = res0;
    if (a.res) {
      const long pix = (long)(y0 + 2 * q + (en >> 4)) * W + x0 + (en & 15);
      const half_t* rp = a.res + (long)tb * a.r_bstride + pix * a.ldr + 32 * m + 16 * eh;
      res0 = *(const half8*)rp;
      res1 = *(const half8*)(rp + 8);
    }
    issue_patch(nvalid, ntb, ny0, nx0, nbuf, g0, g1);
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
    *(half8*)(stg + en * 64 + (((2 * eh) ^ ((en >> 1) & 3)) << 4)) = o0;
    *(half8*)(stg + en * 64 + (((2 * eh + 1) ^ ((en >> 1) & 3)) << 4)) = o1;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    half_t* const yb = (half_t*)a.y + (long)tb * a.y_bstride + ((long)(y0 + 2 * q) * W + x0) * a.ldy + 32 * m;
#pragma unroll
    for (int i = 0; i < 2; ++i) {      // pixels 16 i .. 16 i + 15 of the block = tile row 2 q + i
      const int p = 16 * i + st_p;
      const half8 v = *(const half8*)(stg + p * 64 + ((st_k ^ ((p >> 1) & 3)) << 4));
      *(half8*)(yb + ((long)i * W + st_p) * a.ldy + st_k * 8) = v;
    }
  };

  int vb = blockIdx.x, wb, wy, wx;
  decode(vb, wb, wy, wx);
  int tb = wb, y0 = wy * TH, x0 = wx * TW, ptb = 0, py0 = 0, px0 = 0;
  if (team == 1) issue_patch(true, tb, y0, x0, 0, 0, G_SPLIT);
  else issue_patch(true, tb, y0, x0, 0, G_SPLIT, NPIECES);
  // vmcnt(0) lgkmcnt(0) as the BUILTIN: the compiler's wait-count pass sees it and knows the 36 weight loads are done (with an
  // inline-asm wait it re-waited for them inside the tile loop and drained the patch prefetch and the stores: 9.7 k cycles per tile)
  __builtin_amdgcn_s_waitcnt(0x0070);
  __builtin_amdgcn_s_barrier();

  for (int it = 0;; ++it) {
    ++ntile;
    const bool more = vb + nwg < ntiles;
    int ntb = tb, ny0 = y0, nx0 = x0;
    if (more) {
      step_tile(wb, wy, wx);
      ntb = wb; ny0 = wy * TH; nx0 = wx * TW;
    }
    C64_STAMP(0)   // tile step
    // ---- slot 1
    if (team == 0) {
      k_loop(smem + (it & 1) * PATCH_BYTES);
      C64_STAMP(1)   // K
    } else {
      if (it > 0) epilogue(ptb, py0, px0, more, ntb, ny0, nx0, (it + 1) & 1, 0, G_SPLIT);
      else issue_patch(more, ntb, ny0, nx0, (it + 1) & 1, 0, G_SPLIT);
      C64_STAMP(2)   // epilogue + DMA issue
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    C64_STAMP(3)   // barrier 1
    // ---- slot 2
    if (team == 0) {
      epilogue(tb, y0, x0, more, ntb, ny0, nx0, (it + 1) & 1, G_SPLIT, NPIECES);
      C64_STAMP(2)
    } else {
      k_loop(smem + (it & 1) * PATCH_BYTES);
      C64_STAMP(1)
    }
    // the next patch has landed for this wave: its pieces are older than the two stores of this tile's epilogue (team 1 in
    // its first tile has no stores yet)
    if (team == 1 && it == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
    C64_STAMP(4)   // next patch landed
    __builtin_amdgcn_s_barrier();
    C64_STAMP(5)   // barrier 2
    ptb = tb; py0 = y0; px0 = x0;
    if (!more) break;
    vb += nwg;
    tb = ntb; y0 = ny0; x0 = nx0;
  }
  if (team == 1) epilogue(ptb, py0, px0, false, ptb, py0, px0, 0, 0, 0);   // the last tile of team 1 (no further patch: g0 = g1)
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
