// 3x3 / stride-1 / pad-1 NHWC fp16 convolution for the NARROW layers (Cin = 32, Cout = 32): model.2.m.*,
// cv4.*.1 of YOLOv8s-seg.  Weights-stationary, persistent, HBM-bound by design.
//
// Why a fourth conv kernel.  The im2col kernel gives these layers a 32 ch x 256 px tile: 147 KB of gathered
// activations through the LDS-DMA path per 4.7 MFLOP (31 B/kFLOP, twice the 128x128 tile) -- 75-80 us per layer at
// 160x160, batch 32, against an HBM floor of ~22 us (52 MB in, 52 MB out).  Here
//   * the whole weight matrix (9 taps x 32 rows x 64 B = 18 KB) is loaded into LDS once per workgroup;
//   * a workgroup walks 16 x 16-pixel tiles; per tile the 18 x 20-pixel halo patch (one 32-channel chunk, 23 KB) is
//     the only thing fetched, by 23 LDS-DMA pieces issued during the PREVIOUS tile's nine tap steps into the other
//     patch buffer (out-of-image pixels read a zero page);
//   * nothing in the K loop needs a barrier: weights are static and the patch is complete before the tile starts --
//     one wait + barrier per tile;
//   * four waves, each 32 channels x 4 image rows (MT = 2, NT = 4: 32 fp32 accumulators), two workgroups per CU.
// LDS rows are 64 B with chunk c of row r in slot c ^ (2 * ((r >> 2) & 1)); patch row pitch 20 pixels (see
// conv3x3_wide.hip for both).  Epilogue: conv_epilogue_fast (common.h) for interior tiles.
#include <stdlib.h>

#include "common.h"

namespace m355 {
namespace {

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}
__device__ __forceinline__ void glds4(const void* gsrc, void* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_dst, 4, 0, 0);
}

constexpr int TS = 16, TH = 16;
constexpr int PP = 20, PH = TH + 2;
constexpr int PROWS = 368;                       // 18 x 20 patch pixels + the tail of the last 16-row DMA piece
constexpr int ROWB = 64;
constexpr int PATCH_BYTES = PROWS * ROWB;        // 23552
constexpr int BCH = 32, MT = 2, NT = 4;
constexpr int WTAP = BCH * ROWB;                 // 2048 bytes of weights per tap
constexpr int LDS_BYTES = 2 * PATCH_BYTES + 9 * WTAP + BCH * 4;   // 65,664

__global__ __launch_bounds__(256, 2) void conv3x3_c32_kernel(const ConvArgs a, int tiles_x, int tiles_y, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const wbase = smem + 2 * PATCH_BYTES;
  float* const sbias = (float*)(wbase + 9 * WTAP);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lrow = lane >> 2, lslot = lane & 3;
  const int l15 = lane & 15, g = lane >> 4;
  const int H = a.Hi, W = a.Wi;

  // ---- one-time loads: bias, the 18 weight pieces (tap t, rows 16 h .. 16 h + 15), the first tile's patch
  if (wave == 0 && lane < BCH) glds4(a.bias + lane, sbias);
  for (int pc = wave; pc < 18; pc += 4) {
    const int tap = pc >> 1, R = (pc & 1) * 16 + lrow;          // LDS row R of the tap = MFMA row
    const int mt = R >> 4, r = R & 15;
    const int chl = (r >> 2) * 8 + mt * 4 + (r & 3);            // lane group g ends up with channels 8 g .. 8 g + 7
    const int cc = lslot ^ (((R >> 2) & 1) << 1);
    glds16(a.w + (long)chl * a.Kpad + tap * 32 + cc * 8, wbase + tap * WTAP + (pc & 1) * 1024);
  }

  int tb, ty0, tx0;
  auto decode = [&](int vb) __attribute__((always_inline)) {
    const int xcd = vb & 7, q = ntiles >> 3, r = ntiles & 7;
    const int L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (vb >> 3);
    const int tx = L % tiles_x;
    int rest = L / tiles_x;
    const int ty = rest % tiles_y;
    tb = rest / tiles_y;
    ty0 = ty * TH;
    tx0 = tx * TS;
  };
  // piece j (patch rows 16 j .. 16 j + 15) of the tile (tb, ty0, tx0) into patch buffer `buf`
  auto issue_patch_piece = [&](int buf, int j) __attribute__((always_inline)) {
    const int r = 16 * j + lrow;
    const int py = (r * 205) >> 12;                               // r / 20 for r < 1024
    const int px = r - py * PP;
    const int iy = ty0 - 1 + py, ix = tx0 - 1 + px;
    const bool ok = px < PH && py < PH && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
    const int cc = lslot ^ (((r >> 2) & 1) << 1);
    const half_t* src = ok ? a.x + (long)tb * a.x_bstride + ((long)iy * W + ix) * a.ldx + cc * 8 : a.zero;
    glds16(src, smem + buf * PATCH_BYTES + j * 1024);
  };

  int vb = blockIdx.x;
  decode(vb);
#pragma unroll
  for (int i = 0; i < 6; ++i) issue_patch_piece(0, (4 * i + wave) < 23 ? 4 * i + wave : 22);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  // ---- fragment addressing
  const int fsw = (g ^ (((l15 >> 2) & 1) << 1)) << 4;
  const int aoff = l15 * ROWB + fsw;                               // + tap * WTAP + mt * 1024
  const int pb = (wave * NT) * PP + l15;                           // patch row of (image row 0 of this wave, x = l15)
  const int g16 = g << 4;
  float4v bv[1][2];
  bv[0][0] = *(const float4v*)(sbias + g * 8);
  bv[0][1] = *(const float4v*)(sbias + g * 8 + 4);

  int buf = 0;
  for (;;) {
    const int cb = tb, cy0 = ty0, cx0 = tx0;
    const int nvb = vb + gridDim.x;
    const bool has_next = nvb < ntiles;
    if (has_next) decode(nvb);                                     // the DMA target is now the next tile
    float4v acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = float4v{0.f, 0.f, 0.f, 0.f};
    const char* pbuf = smem + buf * PATCH_BYTES;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int kh = tap / 3, kw = tap - 3 * kh;
      const int p = pb + kh * PP + kw;
      const int be = (p << 6) + (g16 ^ ((p & 4) << 3));
      half8 af[MT], bf[NT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) af[mt] = *(const half8*)(wbase + tap * WTAP + aoff + mt * 1024);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) bf[nt] = *(const half8*)(pbuf + (be ^ ((nt & 1) << 5)) + nt * (PP * ROWB));
      if (tap < 6 && has_next) issue_patch_piece(buf ^ 1, (4 * tap + wave) < 23 ? 4 * tap + wave : 22);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[mt], bf[nt], acc[mt][nt], 0, 0, 0);
    }
    // the next tile's patch has landed (this wave's pieces) and every wave is done reading this tile's patch
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // ---- epilogue
    const int iy0 = cy0 + wave * NT;
    if (iy0 + NT <= H && cx0 + TS <= W && !(a.dbg & 256)) {
      const long pix0 = (long)iy0 * W + cx0 + l15;
      half_t* yp = (half_t*)a.y + (long)cb * a.y_bstride + pix0 * a.ldy + g * 8;
      const long ystep = (long)W * a.ldy;
      if (a.res) {
        const half_t* rp = a.res + (long)cb * a.r_bstride + pix0 * a.ldr + g * 8;
        const long rstep = (long)W * a.ldr;
        if (a.act) conv_epilogue_fast<MT, NT, true, true>(acc, bv, yp, ystep, rp, rstep);
        else conv_epilogue_fast<MT, NT, false, true>(acc, bv, yp, ystep, rp, rstep);
      } else {
        if (a.act) conv_epilogue_fast<MT, NT, true, false>(acc, bv, yp, ystep, nullptr, 0);
        else conv_epilogue_fast<MT, NT, false, false>(acc, bv, yp, ystep, nullptr, 0);
      }
    } else {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int iy = iy0 + nt, ix = cx0 + l15;
        if (iy >= H || ix >= W) continue;
        const long pix = (long)iy * W + ix;
        float v[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          v[j] = acc[0][nt][j] + bv[0][0][j];
          v[4 + j] = acc[1][nt][j] + bv[0][1][j];
        }
        if (a.act) {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = m355_silu(v[j]);
        }
        if (a.res) {
          const half8 rv = *(const half8*)(a.res + (long)cb * a.r_bstride + pix * a.ldr + g * 8);
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] += (float)rv[j];
        }
        half8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = m355_to_half(v[j]);
        *(half8*)((half_t*)a.y + (long)cb * a.y_bstride + pix * a.ldy + g * 8) = o;
      }
    }
    if (!has_next) break;
    vb = nvb;
    buf ^= 1;
  }
}

}  // namespace

// Eligibility: 3x3 stride 1 pad 1, Cin = Cout = 32, fp16 output, 16 x 16 tiles waste at most 30 % of the pixels.
bool conv3x3_c32_ok(const ConvArgs& a) {
  if (a.ksize != 3 || a.stride != 1 || a.pad != 1 || a.out_f32 || a.convt_co > 0 || a.tmode) return false;
  if (a.Cin != 32 || a.Cout != 32 || a.ldx % 8 || a.ldy % 8 || a.Kpad < 288) return false;
  if (a.Ho != a.Hi || a.Wo != a.Wi) return false;
  const long covered = (long)((a.Hi + TH - 1) / TH) * TH * ((a.Wi + TS - 1) / TS) * TS;
  // covered / real pixels allowed, in tenths (M355_C32_WASTE): 3.0 since round 3 (was 1.3) -- the 32 -> 32 convs of the
  // smaller head levels (40 x 40: 1.44, 20 x 20: 2.56) are latency-bound launches of a few MFLOP per CU, where empty tile area
  // costs less than the im2col kernel's prologue: 16.5 -> 8.6 us and 16.3 -> 7.8 us at batch 32
  static const int waste = getenv("M355_C32_WASTE") ? atoi(getenv("M355_C32_WASTE")) : 30;
  return covered * 10 <= (long)a.Hi * a.Wi * waste;
}

int launch_conv3x3_c32(const ConvArgs& a, hipStream_t s) {
  if (!conv3x3_c32_ok(a) || !conv_rows_covered(a, 32)) return -1;
  const int tiles_x = (a.Wi + TS - 1) / TS, tiles_y = (a.Hi + TH - 1) / TH;
  const int B = a.M / (a.Ho * a.Wo);
  const int ntiles = B * tiles_y * tiles_x;
  static int slots = 0;
  if (!slots) {
    hipError_t e = hipFuncSetAttribute((const void*)conv3x3_c32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
      return -2;
    const char* ev = getenv("M355_C32_SLOTS");
    slots = ev ? atoi(ev) : 2 * cus;
    if (slots < 8) slots = 8;
    slots &= ~7;   // the XCD-aware tile order needs gridDim.x % 8 == 0 whenever a block walks more than one tile
  }
  const int grid = ntiles <= slots ? ntiles : slots;
  hipLaunchKernelGGL(conv3x3_c32_kernel, dim3(grid), dim3(256), LDS_BYTES, s, a, tiles_x, tiles_y, ntiles);
  return (int)hipGetLastError();
}

}  // namespace m355
