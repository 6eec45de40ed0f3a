// 3x3 / stride-1 / pad-1 NHWC fp16 convolution with an LDS-STAGED HALO TILE (gfx950, MFMA, fp32 acc).
//
// Why a second conv kernel: the im2col implicit GEMM (conv_igemm.hip) pulls every input pixel through
// the LDS-DMA path nine times (once per tap).  Measured on MI355X the aggregate L2->LDS intake tops out
// near 12.8 TB/s (~24 B/clk/CU) and the 3x3 layers sit exactly on that line (profiles/r01*).  Here one
// workgroup owns a 16x16 output-pixel tile x BCH channels; per 64-channel input chunk the 18x18 halo
// patch is staged ONCE (41 KB) and the nine taps read shifted windows of it from LDS; only the weights
// (BCH x 64 halves per tap) stream per step.  Intake per FLOP drops 3.2x (128 ch) to 3.9x (64 ch).
//
// Same GEMM orientation / fragment maps / epilogue as conv_igemm.hip:
//   D[channel][pixel] += W[channel][tap*Cin + c*64 + k] * patch[pixel + tap][k]
// 8 waves (512 threads); wave (wch, wpx) owns MT*16 channels x NT rows of the 16x16 tile; the 16 lanes
// of an MFMA column block are 16 consecutive x positions of one row, so a B-fragment read touches 16
// consecutive patch pixels: with the 16-byte chunk XOR-swizzled by (patch pixel & 7) it is bank-conflict
// free for every tap shift.  Image borders: out-of-image patch slots are zero-filled once with ds_write
// and the LDS-DMA simply skips those lanes (EXEC-masked), so no zero page is needed.
//
// Pipeline (one barrier per step, placed mid-step, fragments double-buffered in registers):
//   top : ds_read ks=1(step s)            -> 16 MFMA ks=0(s)
//   mid : counted vmcnt + barrier         -> ds_read ks=0(s+1); LDS-DMA weights(s+3) into a 4-deep ring,
//         one patch piece of chunk c+1    -> 16 MFMA ks=1(s)
#include <stdlib.h>

#include "common.h"

namespace m355 {
namespace {

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

__device__ __forceinline__ float silu_f(float v) { return m355_silu(v); }

constexpr int TS = 16;            // output tile is TH rows x TS (16) pixels
constexpr int PW = TS + 2;        // patch width (18)
constexpr int ROWB = 128;         // one LDS row = 64 halves = one pixel's (or weight row's) 64-deep K chunk

// TH = tile rows (16 with 8 waves, 8 with 4 waves): NWAVES = WCH * WPX, NT * WPX = TH.
template <int MT, int NT, int WCH, int WPX>
__global__ __launch_bounds__(WCH * WPX * 64, 2) void conv3x3_halo_kernel(const ConvArgs a, int tiles_x,
                                                                         int tiles_y, int nchunks, int npatch) {
  constexpr int NWAVES = WCH * WPX;
  constexpr int TH = NT * WPX;
  constexpr int PH = TH + 2;
  constexpr int PGROUPS = (PH * PW + 7) / 8;          // patch pixels in groups of 8 LDS rows
  constexpr int PATCH_BYTES = PGROUPS * 8 * ROWB;
  constexpr int P_IT = (PGROUPS + NWAVES - 1) / NWAVES;  // patch row groups per wave
  static_assert(NWAVES == 8 || NWAVES == 4, "4 or 8 waves");
  constexpr int BCH = WCH * MT * 16;
  constexpr int WBUF = BCH * ROWB;
  constexpr int NWB = (WCH * WPX == 8) ? 4 : 2;  // weight ring depth: a stage is in flight for NWB-1 steps
  constexpr int W_IT = BCH / (NWAVES * 8);  // weight LDS-DMA instructions per wave per step
  static_assert(W_IT >= 1, "channel tile too small");
  static_assert(P_IT <= 9, "one patch piece per tap step");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const wbase = smem + npatch * PATCH_BYTES;

  unsigned long long st0 = 0, st1 = 0, st2 = 0, rt0 = 0;
  if (a.stamps) {
    st0 = __builtin_amdgcn_s_memtime();
    rt0 = __builtin_amdgcn_s_memrealtime();
  }
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lrow = lane >> 3;
  const int cc = (lane & 7) ^ lrow;  // K-chunk column this lane fetches (source-side swizzle)

  // ---- block -> tile (XCD-aware: blocks b, b+8, ... share an L2; channel tiles fastest, then x, y, image)
  const int tiles_ch = (a.Cout + BCH - 1) / BCH;
  const int nwg = gridDim.x;
  int L;
  {
    const int orig = blockIdx.x, xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
  }
  const int tile_ch = L % tiles_ch;
  int rest = L / tiles_ch;
  const int tx = rest % tiles_x;
  rest /= tiles_x;
  const int ty = rest % tiles_y;
  const int b = rest / tiles_y;
  const int ch_base = tile_ch * BCH;
  const int y0 = ty * TH, x0 = tx * TS;
  const int H = a.Hi, W = a.Wi;
  const half_t* const xb = a.x + (long)b * a.x_bstride;

  // ---- patch loader state: this wave owns row groups j = wave + 8*i; lane = (row 8j + lane/8, slot lane%8)
  int poff[P_IT];
  unsigned pok = 0;
#pragma unroll
  for (int i = 0; i < P_IT; ++i) {
    const int j = wave + NWAVES * i;
    const int p = 8 * j + lrow;
    const int py = p / PW, px = p - py * PW;
    const int iy = y0 - 1 + py, ix = x0 - 1 + px;
    const bool in_patch = j < PGROUPS && p < PH * PW;
    const bool ok = in_patch && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
    poff[i] = (iy * W + ix) * a.ldx + cc * 8;
    if (ok) pok |= 1u << i;
    if (j < PGROUPS && !ok) {  // border / padding slot: stays zero for every chunk
      const float4v z = {0.f, 0.f, 0.f, 0.f};
      for (int pb = 0; pb < npatch; ++pb) *(float4v*)(smem + pb * PATCH_BYTES + p * ROWB + (lane & 7) * 16) = z;
    }
  }
  auto issue_patch_piece = [&](int chunk, int i) {
    if ((pok >> i) & 1u)
      glds16(xb + poff[i] + chunk * 64, smem + (chunk & (npatch - 1)) * PATCH_BYTES + (wave + NWAVES * i) * 1024);
  };

  // ---- weight loader state: LDS row R (MFMA-tile order) <- permuted source channel (see conv_igemm.hip)
  const half_t* wrow[W_IT];
#pragma unroll
  for (int i = 0; i < W_IT; ++i) {
    const int R = i * NWAVES * 8 + wave * 8 + lrow;
    const int blk = R / (MT * 16), Rl = R % (MT * 16);
    const int mt = Rl >> 4, r = Rl & 15;
    const int chl = (MT >= 2) ? ((mt >> 1) * 32 + (r >> 2) * 8 + (mt & 1) * 4 + (r & 3)) : r;
    wrow[i] = a.w + (long)(ch_base + blk * MT * 16 + chl) * a.Kpad + cc * 8;
  }
  auto issue_weights = [&](int chunk, int tap, int buf) {
    const int koff = tap * a.Cin + chunk * 64;
#pragma unroll
    for (int i = 0; i < W_IT; ++i) glds16(wrow[i] + koff, wbase + buf * WBUF + (i * NWAVES * 8 + wave * 8) * ROWB);
  };

  // ---- fragment addressing
  const int wch = wave / WPX, wpx = wave % WPX;
  const int l15 = lane & 15, g = lane >> 4;
  int aoff[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int r = wch * MT * 16 + mt * 16 + l15;
    aoff[mt] = r * ROWB + ((g ^ (r & 7)) << 4);
  }
  int pbase[NT];  // patch pixel index of (tile row, x = l15) at tap (0,0)
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) pbase[nt] = (wpx * NT + nt) * PW + l15;
  auto baddr = [&](int nt, int chunk, int tap) -> int {
    const int kh = (tap * 11) >> 5, kw = tap - 3 * kh;
    const int p = pbase[nt] + kh * PW + kw;
    return (chunk & (npatch - 1)) * PATCH_BYTES + (p << 7) + ((g ^ (p & 7)) << 4);
  };

  float4v acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = float4v{0.f, 0.f, 0.f, 0.f};
  half8 af0[MT], bf0[NT], af1[MT], bf1[NT];
  int bcur[NT];

  // ---- prologue: patch of chunk 0, weights of steps 0 and 1
#pragma unroll
  for (int i = 0; i < P_IT; ++i) issue_patch_piece(0, i);
#pragma unroll
  for (int i = 0; i < NWB; ++i) issue_weights(0, i, i);  // steps 0..NWB-1 are taps of chunk 0 (NWB <= 9)
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((NWB - 1) * W_IT) : "memory");  // stage 0 (+patch) landed
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) bcur[nt] = baddr(nt, 0, 0);
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) af0[mt] = *(const half8*)(wbase + aoff[mt]);
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) bf0[nt] = *(const half8*)(smem + bcur[nt]);

  // Fast epilogue (common.h) when this wave's rows x 16 columns x 64 channels are all inside the tensor: the bias is
  // fetched here, before the K loop, and waits in registers.
  const bool fast = y0 + wpx * NT + NT <= H && x0 + TS <= W && ch_base + wch * MT * 16 + MT * 16 <= a.Cout && !(a.dbg & (12 | 256));
  float4v bv[MT / 2][2];
  if (fast) {
    const float* bp = a.bias + ch_base + wch * MT * 16 + g * 8;
#pragma unroll
    for (int sg = 0; sg < MT / 2; ++sg) {
      bv[sg][0] = *(const float4v*)(bp + sg * 32);
      bv[sg][1] = *(const float4v*)(bp + sg * 32 + 4);
    }
  }
  if (a.stamps) st1 = __builtin_amdgcn_s_memtime();
  const int nsteps = nchunks * 9;
  int chunk = 0, tap = 0;  // of step s
  for (int s = 0; s < nsteps; ++s) {
    const char* wb = wbase + (s & (NWB - 1)) * WBUF;
    const char* wn = wbase + ((s + 1) & (NWB - 1)) * WBUF;
    // half 1: 8 MFMA ks=0 | ds_read ks=1 fragments | 8 MFMA ks=0.  (Reads are placed BETWEEN MFMA groups
    // because hipcc waits lgkmcnt(0) at the first MFMA after a loop back-edge / branchy region: with the
    // reads issued half a group earlier they have always landed by then.)
#pragma unroll
    for (int mt = 0; mt < MT / 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af0[mt], bf0[nt], acc[mt][nt], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) af1[mt] = *(const half8*)(wb + (aoff[mt] ^ 64));
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bf1[nt] = *(const half8*)(smem + (bcur[nt] ^ 64));
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mt = MT / 2; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af0[mt], bf0[nt], acc[mt][nt], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    // weights(s+1) and every older LDS-DMA must have landed; the NWB-2 younger weight stages may fly
    if (s + NWB - 1 < nsteps)
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((NWB - 2) * W_IT) : "memory");
    else
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // (chunk, tap) of step s+1 and of step s+NWB (whose weights go into the buffer step s just released)
    int c1 = chunk, t1 = tap + 1;
    if (t1 == 9) { t1 = 0; ++c1; }
    int cN = chunk, tN = tap + NWB;
    if (tN >= 9) { tN -= 9; ++cN; }
    // half 2: 8 MFMA ks=1 | LDS-DMA issue for step s+2 (+ one patch piece of the next chunk), ds_read of the
    // ks=0 fragments of step s+1 (past the last step they read stale bytes that are never used) | 8 MFMA ks=1.
    // Everything between the MFMA groups runs in the shadow of the first group's execution.
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mt = 0; mt < MT / 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af1[mt], bf1[nt], acc[mt][nt], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (s + NWB < nsteps && !(a.dbg & 2)) issue_weights(cN, tN, s & (NWB - 1));  // dbg: timing ablations only
    if (tap < P_IT && chunk + 1 < nchunks && !(a.dbg & 1)) {
#pragma unroll
      for (int i = 0; i < P_IT; ++i)
        if (i == tap) issue_patch_piece(chunk + 1, i);
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bcur[nt] = baddr(nt, c1, t1);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) af0[mt] = *(const half8*)(wn + aoff[mt]);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bf0[nt] = *(const half8*)(smem + bcur[nt]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mt = MT / 2; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af1[mt], bf1[nt], acc[mt][nt], 0, 0, 0);
    chunk = c1;
    tap = t1;
  }

  if (a.stamps) st2 = __builtin_amdgcn_s_memtime();
  // ---- epilogue (bias, SiLU, residual, fp16 pack, 16-byte stores at a channel offset)
  if (fast) {
    const long pix0 = (long)(y0 + wpx * NT) * W + x0 + l15;
    const int cho = ch_base + wch * MT * 16 + g * 8;
    half_t* yp = (half_t*)a.y + (long)b * a.y_bstride + pix0 * a.ldy + cho;
    const long ystep = (long)W * a.ldy;
    if (a.res) {
      const half_t* rp = a.res + (long)b * a.r_bstride + pix0 * a.ldr + cho;
      const long rstep = (long)W * a.ldr;
      if (a.act) conv_epilogue_fast<MT, NT, true, true>(acc, bv, yp, ystep, rp, rstep);
      else conv_epilogue_fast<MT, NT, false, true>(acc, bv, yp, ystep, rp, rstep);
    } else {
      if (a.act) conv_epilogue_fast<MT, NT, true, false>(acc, bv, yp, ystep, nullptr, 0);
      else conv_epilogue_fast<MT, NT, false, false>(acc, bv, yp, ystep, nullptr, 0);
    }
    if (a.stamps && tid == 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const unsigned long long st3 = __builtin_amdgcn_s_memtime();
      unsigned long long* o = a.stamps + (long)blockIdx.x * 8;
      o[0] = st0; o[1] = st1; o[2] = st2; o[3] = st3; o[4] = rt0; o[5] = __builtin_amdgcn_s_memrealtime();
    }
    return;
  }
  constexpr int GROUPS = (MT >= 2) ? MT / 2 : 1;
  constexpr int GW = (MT >= 2) ? 8 : 4;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int iy = y0 + wpx * NT + nt, ix = x0 + l15;
    if (iy >= H || ix >= W) continue;
    const long pix = (long)iy * W + ix;
#pragma unroll
    for (int s = 0; s < GROUPS; ++s) {
      const int ch0 = ch_base + wch * MT * 16 + ((MT >= 2) ? (s * 32 + g * 8) : (g * 4));
      if (ch0 >= a.Cout) continue;
      float v[GW];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (MT >= 2) {
          v[j] = acc[2 * s][nt][j];
          v[4 + j] = acc[2 * s + 1][nt][j];
        } else {
          v[j] = acc[0][nt][j];
        }
      }
#pragma unroll
      for (int j = 0; j < GW; ++j) v[j] += a.bias[ch0 + j];
      if (a.act && !(a.dbg & 4)) {
#pragma unroll
        for (int j = 0; j < GW; ++j) v[j] = silu_f(v[j]);
      }
      if (a.res) {
        const half_t* rp = a.res + (long)b * a.r_bstride + pix * a.ldr + ch0;
        if (GW == 8) {
          const half8 rv = *(const half8*)rp;
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] += (float)rv[j];
        } else {
          const half4 rv = *(const half4*)rp;
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] += (float)rv[j];
        }
      }
      half_t* yp = (half_t*)a.y + (long)b * a.y_bstride + pix * a.ldy + ch0;
      if ((a.dbg & 8) && v[0] != 123.f) continue;  // dbg: no stores
      if (GW == 8) {
        half8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = m355_to_half(v[j]);
        *(half8*)yp = o;
      } else {
        half4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = m355_to_half(v[j]);
        *(half4*)yp = o;
      }
    }
  }
  if (a.stamps && tid == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long st3 = __builtin_amdgcn_s_memtime();
    unsigned long long* o = a.stamps + (long)blockIdx.x * 8;  // [4..5]: 100 MHz realtime -> in-kernel clock
    o[0] = st0; o[1] = st1; o[2] = st2; o[3] = st3; o[4] = rt0; o[5] = __builtin_amdgcn_s_memrealtime();
  }
}

template <int MT, int NT, int WCH, int WPX>
int launch_halo_variant(const ConvArgs& a, hipStream_t s) {
  constexpr int BCH = WCH * MT * 16;
  constexpr int TH = NT * WPX;
  constexpr int PGROUPS = ((TH + 2) * PW + 7) / 8;
  const int tiles_x = (a.Wi + TS - 1) / TS, tiles_y = (a.Hi + TH - 1) / TH;
  const int tiles_ch = (a.Cout + BCH - 1) / BCH;
  const int nchunks = a.Cin / 64;
  const int npatch = nchunks > 1 ? 2 : 1;
  const int B = a.M / (a.Ho * a.Wo);
  if (!conv_rows_covered(a, BCH)) return -1;
  const int lds = npatch * PGROUPS * 8 * ROWB + ((WCH * WPX == 8) ? 4 : 2) * BCH * ROWB;
  auto k = conv3x3_halo_kernel<MT, NT, WCH, WPX>;
  if (lds > 65536) {
    hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(k, dim3(B * tiles_y * tiles_x * tiles_ch), dim3(WCH * WPX * 64), lds, s, a, tiles_x, tiles_y,
                     nchunks, npatch);
  return (int)hipGetLastError();
}

}  // namespace

// Eligibility: 3x3 stride 1 pad 1, fp16 output, Cin a multiple of 64, Cout >= 64, and the 16-wide tiling
// wastes at most 30 % of the computed pixels.
bool conv3x3_halo_ok(const ConvArgs& a) {
  if (a.ksize != 3 || a.stride != 1 || a.pad != 1 || a.out_f32 || a.convt_co > 0) return false;
  if (a.Cin % 64 || a.Cout < 64 || a.Cout % 8 || a.ldx % 8 || a.ldy % 8) return false;
  if (a.Ho != a.Hi || a.Wo != a.Wi) return false;
  const long covered = (long)((a.Hi + 7) / 8) * 8 * ((a.Wi + TS - 1) / TS) * TS;
  return covered * 10 <= (long)a.Hi * a.Wi * 13;
}

// variant: 0 = auto, 1 = 8 waves / 16x16 px, 2 = 4 waves / 8x16 px
int launch_conv3x3_halo(const ConvArgs& a0, int variant, hipStream_t s) {
  ConvArgs a = a0;
  if (knobs().no_fast_epi) a.dbg |= 256;
  if (variant == TILE_SLAB - TILE_HALO) return launch_conv3x3_slab(a, s);
  if (!conv3x3_halo_ok(a)) return -1;
  if (variant >= TILE_M32 - TILE_HALO && variant <= TILE_M32_64x8 - TILE_HALO) return launch_conv3x3_m32(a, variant - (TILE_M32 - TILE_HALO), s);
  // by shape (measured at batch 32, tools/conv3_sweep.py): the wide kernel where its 16x16-pixel tiles fit (80x80 and
  // larger maps, Cout >= 128: equal to the 32x32x16 kernel there); the 32x32x16 kernel on the smaller maps (128 -> 128 at
  // 40x40: 25.8 -> 21.7 us, 256 -> 224: 79.7 -> 68.8 us, 64 -> 64: 10.0 -> 8.7 us); the 8-row halo kernel for 64-channel
  // layers on large maps (24.6 us against 26.0)
  if (variant == 0 && !knobs().no_m32 && conv3x3_m32_ok(a) && !(conv3x3_wide_ok(a) && !knobs().no_wide) &&
      (a.Cout > 64 || a.Hi * a.Wi <= 1600))
    return launch_conv3x3_m32(a, 0, s);
  if (variant == 3 || (variant == 0 && conv3x3_wide_ok(a) && !knobs().no_wide)) return launch_conv3x3_wide(a, s);
  if (variant >= 5) return -1;   // ids 21-24 were the lean halo template (measured equal to the K-64 kernels below; removed in round 3)
  if (variant == 0) {
    variant = knobs().halo_variant;  // measured: the 4-wave variant (two blocks per CU) wins on every layer
    if (variant != 1 && variant != 2) variant = 2;
  }
  if (variant == 1) {
    if (a.Cout > 64) return launch_halo_variant<4, 4, 2, 4>(a, s);  // 128 ch x 256 px, wave 64 ch x 64 px
    return launch_halo_variant<4, 2, 1, 8>(a, s);                   //  64 ch x 256 px, wave 64 ch x 32 px
  }
  if (a.Cout > 64) return launch_halo_variant<4, 4, 2, 2>(a, s);    // 128 ch x 128 px, wave 64 ch x 64 px
  return launch_halo_variant<4, 2, 1, 4>(a, s);                     //  64 ch x 128 px, wave 64 ch x 32 px
}

}  // namespace m355
