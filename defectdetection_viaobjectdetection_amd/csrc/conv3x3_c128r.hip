// 3x3 / stride-1 / pad-1 NHWC fp16 convolution with Cin = Cout = 128, WEIGHTS IN REGISTERS, K split over the two waves of
// a SIMD (gfx950, v_mfma_f32_32x32x16_f16).
//
// Replaces: the Conv+BN+SiLU 3x3 layers of the 128-channel C2f bottlenecks on the 40 x 40 maps
// (YOLOv8s-seg model.6.m.*, model.12.m.*, model.18.m.*) and the stride-16 class branch (model.22.cv3.1.1) that upstream
// reaches through torch.nn.functional.conv2d (call site: /root/reference/BscanBased/yolo8_seg_predict.py:8).
//
// Why.  On the 32x32x16 halo kernel (conv3x3_m32.hip) these nine launches ran 24-27 us each for 15.1 GFLOP (0.24 of the MFMA
// peak): 480 tiles on 512 block slots = one tile per block, so nothing amortised the prologue, every tile re-streamed the
// 295 KB weight matrix through LDS-DMA, and 16-pixel-wide tiles waste a sixth of a 40-pixel-wide map.  conv3x3_c64r.hip
// showed the other form for 64 channels: weights in registers, persistent blocks, only the activation patch in LDS.  A
// 128 x 1152 matrix is 288 VGPRs per 32-channel block -- too many for one wave at two waves per SIMD -- so the K axis is
// SPLIT BY INPUT-CHANNEL HALF over the two waves that share a SIMD: wave (m, kh) holds the 36 fragments of channel block m
// for input channels 64 kh .. 64 kh + 63 (144 VGPRs) and accumulates a PARTIAL sum for both 32-pixel blocks of the tile;
// the partners exchange one block each through LDS (fp32), add, and each finishes one block (SiLU, + residual, fp16,
// store).  Tile = 8 x 8 pixels (40 = 5 x 8: no column waste), pixel block = 4 rows x 8 columns.
//
// LDS image: one 256-byte row per patch pixel (10 x 10 patch, 128 channels), 16-byte chunk index XOR-ed with
// ((patch row & 3) << 2 | (patch column & 3)) on the DMA source side and on the reads: the 16 lanes of a ds_read_b128
// service group (four rows x four columns of the block, shifted by the tap) then read 16 distinct bank groups.
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace m355 {
namespace {

typedef float float16v __attribute__((ext_vector_type(16)));

constexpr int TH = 8, TW = 8, PP = 10, ROWB = 256;
constexpr int PROWS = (TH + 2) * PP;                 // 100 patch pixels
constexpr int NPIECES = PROWS / 4;                   // 25 DMA pieces of 4 rows
constexpr int PATCH_BYTES = NPIECES * 1024;          // 25600
constexpr int NBUF = 3;
constexpr int NWAVES = 8;
constexpr int P_IT = (NPIECES + NWAVES - 1) / NWAVES;   // 4
constexpr int XCH_OFF = NBUF * PATCH_BYTES;          // partial-sum exchange: 8 waves x 64 lanes x 16 floats
constexpr int STG_OFF = XCH_OFF + NWAVES * 4096;     // output staging: 8 waves x 32 pixels x 64 bytes
constexpr int BIAS_OFF = STG_OFF + NWAVES * 2048;
constexpr int W8_OFF = BIAS_OFF + 512;               // the four tap-8 weight fragments of every wave (lane-linear: conflict free)
constexpr int LDS_BYTES = W8_OFF + NWAVES * 4096;    // 159232

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, int voff, int soff, char* lds) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds, 16, voff, soff, 0, 0);
}
__device__ __forceinline__ int row_plain(int rho) { return 16 * ((rho >> 2) & 1) + 4 * (rho >> 3) + (rho & 3); }

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

__global__ __launch_bounds__(512, 2) void conv3x3_c128r_kernel(const ConvArgs a, int tiles_x, int tiles_y, int ntiles, int sx, int sy,
                                                              int sb) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = a.Hi, W = a.Wi, nwg = gridDim.x;
  const int n = lane & 31, h = lane >> 5;
  const int m = wave & 3, kh_ = wave >> 2;            // channel block, input-channel half (waves w and w + 4 share a SIMD)

  if (tid < 128) ((float*)(smem + BIAS_OFF))[tid] = a.bias[tid];

  // ---- this wave's weights: (tap, slice) fragments of channel block m over input channels 64 kh_ .. + 63
  // (taps 0-7 in 128 VGPRs; the four fragments of tap 8 live in this wave's 4 KB of LDS and are read once per tile -- the
  // register budget of two waves per SIMD is 256 and 144 + 32 accumulators + 32 activation fragments + addressing spilled)
  half8 wv[32];
  if (a.wf) {      // fragment-ordered copy [m][kh][tap][slice]: one coalesced 1 KiB load per fragment
    const half_t* wp = a.wf + (long)(m * 2 + kh_) * 36 * 512 + lane * 8;
#pragma unroll
    for (int f = 0; f < 32; ++f) wv[f] = *(const half8*)(wp + 512 * f);
#pragma unroll
    for (int s = 0; s < 4; ++s) *(half8*)(smem + W8_OFF + wave * 4096 + s * 1024 + lane * 16) = *(const half8*)(wp + 512 * (32 + s));
  } else {
    const half_t* wp = a.w + (long)(32 * m + row_plain(n)) * a.Kpad + 64 * kh_ + 8 * h;
#pragma unroll
    for (int tap = 0; tap < 8; ++tap)
#pragma unroll
      for (int s = 0; s < 4; ++s) wv[4 * tap + s] = *(const half8*)(wp + 128 * tap + 16 * s);
#pragma unroll
    for (int s = 0; s < 4; ++s) *(half8*)(smem + W8_OFF + wave * 4096 + s * 1024 + lane * 16) = *(const half8*)(wp + 128 * 8 + 16 * s);
  }

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

  // ---- patch pieces: wave w owns pieces g = w + 8 i (4 LDS rows each); lane = (row 4 g + lane / 16, chunk slot lane % 16)
  auto issue_patch = [&](int tb, int y0, int x0, int buf) __attribute__((always_inline)) {
    const int origin = (((y0 - 1) * W + (x0 - 1)) * a.ldx) * 2;
    const bool interior = y0 >= 1 && y0 + TH + 1 <= H && x0 >= 1 && x0 + TW + 1 <= W;
    // (piece geometry is rebuilt from mbcnt per issue: kept live across the K loop it is spilled, and a scratch reload waits on
    // vmcnt, i.e. on the DMA issued just before it -- and breaks the counted waits below)
    int ln;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
    const int r0 = 4 * wave + (ln >> 4), slot = ln & 15;
#pragma unroll
    for (int i = 0; i < P_IT; ++i) {
      const int g = wave + NWAVES * i;
      if (g < NPIECES) {
        const int R = r0 + 4 * NWAVES * i;
        const int pr = (R * 6554) >> 16, pc = R - pr * PP;           // R / 10, R % 10 (exact for R < 200)
        const int yy = y0 - 1 + pr, xx = x0 - 1 + pc;
        const bool ok = interior || ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W);
        const int rel = ((pr * W + pc) * a.ldx + ((slot ^ (((pr & 3) << 2) | (pc & 3))) << 3)) * 2;
        dma16(rs_x, ok ? origin + rel : (int)0x80000000, tb * img_stride, smem + buf * PATCH_BYTES + g * 1024);
      }
    }
  };

  // ---- fragment offsets: pixel block pb (tile rows 4 pb .. 4 pb + 3), lane pixel (row n >> 3, column n & 7); the swizzle of
  // patch pixel (row + kh, column + kw) depends on both shifts: one offset per (pixel block, kh & 3 class, kw, slice) would be
  // 72 registers, so the row part of the XOR is applied per tap row (it only flips bits 2-3 of the chunk index)
  const int prow0 = n >> 3, pcol0 = n & 7;
  int offc[2][3];            // byte offset of the pixel's row at tap (0, kw): (4 pb + prow0) * PP + pcol0 + kw
  int swc[3];                // column part of the swizzle at kw
#pragma unroll
  for (int kw = 0; kw < 3; ++kw) {
    swc[kw] = (pcol0 + kw) & 3;
#pragma unroll
    for (int pb = 0; pb < 2; ++pb) offc[pb][kw] = ((4 * pb + prow0) * PP + pcol0 + kw) * ROWB;
  }
  const int chunk0 = 8 * kh_ + h;                    // + 2 s: this lane's 16-byte chunk of K slice s in a pixel row

  char* const xch = smem + XCH_OFF;
  char* const stg = smem + STG_OFF + wave * 2048;
  const int partner = wave ^ 4;

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

  auto run = [&](auto khc) __attribute__((always_inline)) {
  constexpr int KH = decltype(khc)::value;              // = kh_ (compile time: the accumulators are indexed by it)
  for (int it = 0;; ++it) {
    plan(2);
    const char* const pbuf = smem + (it % NBUF) * PATCH_BYTES;
    if (have[2]) issue_patch(tbi[2], ty0[2], tx0[2], (it + 2) % NBUF);
    // ---- K loop over this wave's half of K: 9 taps x 4 slices x 2 pixel blocks; fragments of tap t + 1 under tap t
    float16v acc[2];
    {
      float16v bv;
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const float4v u = *(const float4v*)(smem + BIAS_OFF + (32 * m + 16 * h + 4 * qd) * 4);
        bv[qd * 4 + 0] = u[0]; bv[qd * 4 + 1] = u[1]; bv[qd * 4 + 2] = u[2]; bv[qd * 4 + 3] = u[3];
      }
      // the bias enters once: through the partial sum of the block this wave finishes
      acc[KH] = bv;
      acc[KH ^ 1] = (float16v)0.f;
    }
    // fragments: one set per pixel block; the set of (tap + 1, block) is read right behind the four MFMAs of (tap, block),
    // i.e. four MFMAs (128 cycles) ahead of its use
    half8 fr[2][4];
    auto read_frag = [&](int tap, int pb) __attribute__((always_inline)) {
      const int kh = tap / 3, kw = tap - 3 * kh;
      const int swr = ((4 * pb + prow0 + kh) & 3) << 2;
#pragma unroll
      for (int s = 0; s < 4; ++s)
        fr[pb][s] = *(const half8*)(pbuf + offc[pb][kw] + kh * PP * ROWB + (((chunk0 + 2 * s) ^ (swr | swc[kw])) << 4));
    };
    read_frag(0, 0);
    read_frag(0, 1);
#pragma unroll
    for (int tap = 0; tap < 8; ++tap) {
#pragma unroll
      for (int pb = 0; pb < 2; ++pb) {
#pragma unroll
        for (int s = 0; s < 4; ++s) acc[pb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wv[4 * tap + s], fr[pb][s], acc[pb], 0, 0, 0);
        read_frag(tap + 1, pb);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    {
      half8 w8[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) w8[s] = *(const half8*)(smem + W8_OFF + wave * 4096 + s * 1024 + lane * 16);
#pragma unroll
      for (int pb = 0; pb < 2; ++pb)
#pragma unroll
        for (int s = 0; s < 4; ++s) acc[pb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w8[s], fr[pb][s], acc[pb], 0, 0, 0);
    }
    // (lane-derived values of the epilogue are rebuilt from mbcnt here: kept live across the K loop they are spilled, and a
    // scratch reload waits on vmcnt(0) = drains the patch prefetch and the previous tile's stores)
    int e_lane;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(e_lane));
    const int e_n = e_lane & 31, e_h = e_lane >> 5, e_prow0 = e_n >> 3, e_pcol0 = e_n & 7;
    const int e_st_w0 = e_n * 64 + (((2 * e_h) ^ ((e_n >> 1) & 3)) << 4), e_st_w1 = e_n * 64 + (((2 * e_h + 1) ^ ((e_n >> 1) & 3)) << 4);
    const int e_st_p = e_lane >> 2, e_st_k = e_lane & 3;
    // ---- exchange: give the partner the partial sum of ITS block, take the partner's partial sum of mine
    {
      float* const mine = (float*)(xch + wave * 4096) + e_lane * 4;
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        float4v u;
        u[0] = acc[KH ^ 1][qd * 4 + 0]; u[1] = acc[KH ^ 1][qd * 4 + 1]; u[2] = acc[KH ^ 1][qd * 4 + 2]; u[3] = acc[KH ^ 1][qd * 4 + 3];
        *(float4v*)(mine + qd * 256) = u;                 // [quad][lane][4]: conflict-free 16-byte rows
      }
    }
    // this wave FINISHES pixel block KH of the tile: its residual (accumulator layout) lands under the barrier and the SiLUs
    // (loaded here, not at the top of the tile: eight registers the K loop does not have)
    half8 res0 = (half8)(half_t)0.f, res1 = res0;
    if (a.res) {
      const long pix = (long)(ty0[0] + 4 * KH + e_prow0) * W + tx0[0] + e_pcol0;
      const half_t* rp = a.res + (long)tbi[0] * a.r_bstride + pix * a.ldr + 32 * m + 16 * e_h;
      res0 = *(const half8*)rp;
      res1 = *(const half8*)(rp + 8);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                          // (every wave: its partial sums are in LDS; the patch of this tile is free)
    float16v fin = acc[KH];
    {
#pragma clang fp contract(off)
      const float* const theirs = (const float*)(xch + partner * 4096) + e_lane * 4;
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const float4v u = *(const float4v*)(theirs + qd * 256);
        fin[qd * 4 + 0] = fin[qd * 4 + 0] + u[0]; fin[qd * 4 + 1] = fin[qd * 4 + 1] + u[1];
        fin[qd * 4 + 2] = fin[qd * 4 + 2] + u[2]; fin[qd * 4 + 3] = fin[qd * 4 + 3] + u[3];
      }
    }
    // ---- epilogue of pixel block kh_: SiLU, + residual, fp16, transpose through LDS, 64-byte row segments out
    if (a.act) silu16(fin);
    half8 o0, o1;
    {
#pragma clang fp contract(off)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float v0 = fin[j], v1 = fin[8 + j];
        if (a.res) { v0 = v0 + (float)res0[j]; v1 = v1 + (float)res1[j]; }
        o0[j] = m355_to_half(v0);
        o1[j] = m355_to_half(v1);
      }
    }
    *(half8*)(stg + e_st_w0) = o0;
    *(half8*)(stg + e_st_w1) = o1;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    {
      half_t* const yb = (half_t*)a.y + (long)tbi[0] * a.y_bstride + ((long)(ty0[0] + 4 * KH) * W + tx0[0]) * a.ldy + 32 * m;
#pragma unroll
      for (int i = 0; i < 2; ++i) {      // staged pixels 16 i .. 16 i + 15 = block rows 2 i, 2 i + 1 (8 columns each)
        const int p = 16 * i + e_st_p;
        const half8 v = *(const half8*)(stg + p * 64 + ((e_st_k ^ ((p >> 1) & 3)) << 4));
        *(half8*)(yb + ((long)(p >> 3) * W + (p & 7)) * a.ldy + e_st_k * 8) = v;
      }
    }
    // the patch of tile it + 1 has landed for this wave: everything older than this iteration's own pieces and stores
    if (have[2]) {
      if (wave + 3 * NWAVES < NPIECES) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");   // 4 pieces + 2 stores
      else asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)" ::: "memory");                                // 3 pieces + 2 stores
    } else {
      asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (!have[1]) break;
#pragma unroll
    for (int k = 0; k < 2; ++k) { tbi[k] = tbi[k + 1]; ty0[k] = ty0[k + 1]; tx0[k] = tx0[k + 1]; have[k] = have[k + 1]; }
  }
  };
  if (kh_ == 0) run(std::integral_constant<int, 0>{});
  else run(std::integral_constant<int, 1>{});
}

}  // namespace

// Eligibility: 3x3 / s1 / p1, Cin = Cout = 128, fp16 out, map a multiple of the 8 x 8 tile, 31-bit byte offsets.
bool conv3x3_c128r_ok(const ConvArgs& a) {
  if (a.ksize != 3 || a.stride != 1 || a.pad != 1 || a.out_f32 || a.convt_co > 0 || a.tmode || a.phase || a.csplit || a.w2 || a.dec_preds)
    return false;
  if (a.Cin != 128 || a.Cout != 128 || a.ldx % 8 || a.ldy % 8 || a.Kpad < 1152 || a.Kpad % 8) return false;
  if (a.Ho != a.Hi || a.Wo != a.Wi || a.Hi % TH || a.Wi % TW) return false;
  if (a.res && a.ldr % 8) return false;
  const long nimg = a.Ho * a.Wo > 0 ? a.M / ((long)a.Ho * a.Wo) : 0;
  if (nimg < 1) return false;
  return ((nimg - 1) * a.x_bstride + (long)a.Hi * a.Wi * a.ldx) * 2 < (1L << 31);
}

int launch_conv3x3_c128r(const ConvArgs& a, hipStream_t s) {
  if (!conv3x3_c128r_ok(a) || !conv_rows_covered(a, 128)) return -1;
  const int tiles_x = a.Wi / TW, tiles_y = a.Hi / TH;
  const int B = a.M / (a.Ho * a.Wo);
  const int ntiles = B * tiles_y * tiles_x;
  static int slots = 0;
  if (!slots) {
    hipError_t e = hipFuncSetAttribute((const void*)conv3x3_c128r_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
      return -2;
    slots = cus & ~7;
    if (slots < 8) slots = 8;
  }
  const int grid = ntiles <= slots ? ntiles : slots;
  const int step = grid >> 3;
  const int sx = step % tiles_x, sy = (step / tiles_x) % tiles_y, sb = step / tiles_x / tiles_y;
  hipLaunchKernelGGL(conv3x3_c128r_kernel, dim3(grid), dim3(64 * NWAVES), LDS_BYTES, s, a, tiles_x, tiles_y, ntiles, sx, sy, sb);
  return (int)hipGetLastError();
}

}  // namespace m355
