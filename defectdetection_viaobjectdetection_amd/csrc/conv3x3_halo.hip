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
//   mid : vmcnt(0)+lgkmcnt(0)+barrier     -> ds_read ks=0(s+1); LDS-DMA weights(s+2), one patch piece
//         of chunk c+1                    -> 16 MFMA ks=1(s)
#include "common.h"

namespace m355 {
namespace {

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

__device__ __forceinline__ float silu_f(float v) {
  float e = __builtin_amdgcn_exp2f(v * -1.4426950408889634f);
  return v * __builtin_amdgcn_rcpf(1.0f + e);
}

constexpr int TS = 16;            // output tile is TS x TS pixels
constexpr int PW = TS + 2;        // patch width / height (18)
constexpr int PROWS = 328;        // 18*18 = 324 patch pixels padded to 41 groups of 8 rows
constexpr int PGROUPS = PROWS / 8;
constexpr int ROWB = 128;         // one LDS row = 64 halves = one pixel's (or weight row's) 64-deep K chunk
constexpr int PATCH_BYTES = PROWS * ROWB;
constexpr int P_IT = (PGROUPS + 7) / 8;  // patch row groups per wave (6)

template <int MT, int NT, int WCH, int WPX>
__global__ __launch_bounds__(512, 2) void conv3x3_halo_kernel(const ConvArgs a, int tiles_x, int tiles_y,
                                                              int nchunks, int npatch) {
  static_assert(WCH * WPX == 8 && NT * WPX == TS, "8 waves cover 16 rows");
  constexpr int BCH = WCH * MT * 16;
  constexpr int WBUF = BCH * ROWB;
  constexpr int W_IT = BCH / 64;  // weight LDS-DMA instructions per wave per step (8 waves x 8 rows)
  static_assert(W_IT >= 1, "channel tile too small");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const wbase = smem + npatch * PATCH_BYTES;

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
  const int y0 = ty * TS, x0 = tx * TS;
  const int H = a.Hi, W = a.Wi;
  const half_t* const xb = a.x + (long)b * a.x_bstride;

  // ---- patch loader state: this wave owns row groups j = wave + 8*i; lane = (row 8j + lane/8, slot lane%8)
  int poff[P_IT];
  unsigned pok = 0;
#pragma unroll
  for (int i = 0; i < P_IT; ++i) {
    const int j = wave + 8 * i;
    const int p = 8 * j + lrow;
    const int py = p / PW, px = p - py * PW;
    const int iy = y0 - 1 + py, ix = x0 - 1 + px;
    const bool in_patch = j < PGROUPS && p < PW * PW;
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
      glds16(xb + poff[i] + chunk * 64, smem + (chunk & (npatch - 1)) * PATCH_BYTES + (wave + 8 * i) * 1024);
  };

  // ---- weight loader state: LDS row R (MFMA-tile order) <- permuted source channel (see conv_igemm.hip)
  const half_t* wrow[W_IT];
#pragma unroll
  for (int i = 0; i < W_IT; ++i) {
    const int R = i * 64 + wave * 8 + lrow;
    const int blk = R / (MT * 16), Rl = R % (MT * 16);
    const int mt = Rl >> 4, r = Rl & 15;
    const int chl = (MT >= 2) ? ((mt >> 1) * 32 + (r >> 2) * 8 + (mt & 1) * 4 + (r & 3)) : r;
    wrow[i] = a.w + (long)(ch_base + blk * MT * 16 + chl) * a.Kpad + cc * 8;
  }
  auto issue_weights = [&](int chunk, int tap, int buf) {
    const int koff = tap * a.Cin + chunk * 64;
#pragma unroll
    for (int i = 0; i < W_IT; ++i) glds16(wrow[i] + koff, wbase + buf * WBUF + (i * 64 + wave * 8) * ROWB);
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
  issue_weights(0, 0, 0);
  issue_weights(0, 1, 1);
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(W_IT) : "memory");  // weights(1) may still fly
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) bcur[nt] = baddr(nt, 0, 0);
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) af0[mt] = *(const half8*)(wbase + aoff[mt]);
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) bf0[nt] = *(const half8*)(smem + bcur[nt]);

  const int nsteps = nchunks * 9;
  int chunk = 0, tap = 0;  // of step s
  for (int s = 0; s < nsteps; ++s) {
    const char* wb = wbase + (s & 1) * WBUF;
    const char* wn = wbase + ((s + 1) & 1) * WBUF;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) af1[mt] = *(const half8*)(wb + (aoff[mt] ^ 64));
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bf1[nt] = *(const half8*)(smem + (bcur[nt] ^ 64));
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af0[mt], bf0[nt], acc[mt][nt], 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // (chunk, tap) of steps s+1 and s+2
    int c1 = chunk, t1 = tap + 1;
    if (t1 == 9) { t1 = 0; ++c1; }
    int c2 = c1, t2 = t1 + 1;
    if (t2 == 9) { t2 = 0; ++c2; }
    // ks=0 fragments of step s+1 (past the last step this reads stale bytes that are never used)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bcur[nt] = baddr(nt, c1, t1);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) af0[mt] = *(const half8*)(wn + aoff[mt]);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bf0[nt] = *(const half8*)(smem + bcur[nt]);
    if (s + 2 < nsteps) issue_weights(c2, t2, s & 1);
    if (tap < P_IT && chunk + 1 < nchunks) {
#pragma unroll
      for (int i = 0; i < P_IT; ++i)
        if (i == tap) issue_patch_piece(chunk + 1, i);
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af1[mt], bf1[nt], acc[mt][nt], 0, 0, 0);
    chunk = c1;
    tap = t1;
  }

  // ---- epilogue (bias, SiLU, residual, fp16 pack, 16-byte stores at a channel offset)
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
      if (a.act) {
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
      if (GW == 8) {
        half8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (half_t)v[j];
        *(half8*)yp = o;
      } else {
        half4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (half_t)v[j];
        *(half4*)yp = o;
      }
    }
  }
}

template <int MT, int NT, int WCH, int WPX>
int launch_halo_variant(const ConvArgs& a, hipStream_t s) {
  constexpr int BCH = WCH * MT * 16;
  const int tiles_x = (a.Wi + TS - 1) / TS, tiles_y = (a.Hi + TS - 1) / TS;
  const int tiles_ch = (a.Cout + BCH - 1) / BCH;
  const int nchunks = a.Cin / 64;
  const int npatch = nchunks > 1 ? 2 : 1;
  const int B = a.M / (a.Ho * a.Wo);
  const int lds = npatch * PATCH_BYTES + 2 * BCH * ROWB;
  auto k = conv3x3_halo_kernel<MT, NT, WCH, WPX>;
  if (lds > 65536) {
    hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(k, dim3(B * tiles_y * tiles_x * tiles_ch), dim3(512), lds, s, a, tiles_x, tiles_y, nchunks,
                     npatch);
  return (int)hipGetLastError();
}

}  // namespace

// Eligibility: 3x3 stride 1 pad 1, fp16 output, Cin a multiple of 64, Cout >= 64, and the 16x16 tiling
// wastes at most 30 % of the computed pixels.
bool conv3x3_halo_ok(const ConvArgs& a) {
  if (a.ksize != 3 || a.stride != 1 || a.pad != 1 || a.out_f32 || a.convt_co > 0) return false;
  if (a.Cin % 64 || a.Cout < 64 || a.Cout % 8 || a.ldx % 8 || a.ldy % 8) return false;
  if (a.Ho != a.Hi || a.Wo != a.Wi) return false;
  const long covered = (long)((a.Hi + TS - 1) / TS) * TS * ((a.Wi + TS - 1) / TS) * TS;
  return covered * 10 <= (long)a.Hi * a.Wi * 13;
}

int launch_conv3x3_halo(const ConvArgs& a, hipStream_t s) {
  if (!conv3x3_halo_ok(a)) return -1;
  if (a.Cout > 64) return launch_halo_variant<4, 4, 2, 4>(a, s);  // 128 ch x 256 px, wave 64 ch x 64 px
  return launch_halo_variant<4, 2, 1, 8>(a, s);                   //  64 ch x 256 px, wave 64 ch x 32 px
}

}  // namespace m355
