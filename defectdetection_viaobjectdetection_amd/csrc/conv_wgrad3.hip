// Weight gradient of a 3x3 / stride-1 / pad-1 NHWC fp16 convolution from spatial PATCHES (gfx950, fp32 accumulate).
// Replaces the wgrad half of aten conv2d's autograd for the 3x3 layers (call site
// BscanBased/yolo_seg_train.py:12); conv_wgrad.hip keeps the 1x1, 2x2 and stride-2 layers.
//
//   dW[co][kh][kw][ci] = sum over (b, y, x) of dZ[b, y, x, co] * X[b, y - 1 + kh, x - 1 + kw, ci]
//
// The pixel-axis GEMM of conv_wgrad.hip gathers, for every 64 pixels, a 128-column slice of the im2col matrix: the same
// input pixel enters LDS nine times (once per tap), 32 KB of LDS-DMA per 2.1 MFLOP, and the kernel runs at the DMA latency
// (1.96 us per K step on the 128 -> 128 layer, 0.25 us of MFMA).  Here a block stages an 8 x 16 output tile of dZ and the
// 10 x 18 input patch around it ONCE and reads all nine taps out of that patch -- 39 KB per 18.9 MFLOP (64 x 64 channels).
//
// MFMA v_mfma_f32_32x32x16_f16, D[co][ci] += A[co][px] * B[px][ci]: K = 16 pixels = one row of the tile; a lane needs 8
// consecutive pixels of ONE channel, i.e. a column of the NHWC image: gfx950's transposed LDS read ds_read_b64_tr_b16
// (4 rows x 16 columns per 16 lanes), two reads per fragment.  The A fragment (dZ row) is read once per tile row and used
// by nine MFMAs whose B fragments are the same 16 pixels shifted by the tap: row (c + kh) of the patch, column kw.
//
// LDS rows are pixels.  128-byte rows (64 channels) XOR the 16-byte chunk index with 4 * ((row >> 1) & 1): a transposed
// read touches rows r .. r + 3 x 64 bytes per half wave, and any four consecutive rows then cover the four 64-byte
// quarters of the bank space -- conflict free for EVERY row offset, which the tap shifts need.  64-byte rows need nothing.
// The swizzle is applied on the source side of the LDS-DMA (which chunk a lane fetches) and on the read side.
//
// Block = 32 NCO output channels x 32 NCI input channels x 9 taps, four waves: one 32 x 32 channel block per wave and, when
// the tile has fewer than four of them, the tile's rows split across the spare waves (partials added through LDS in wave
// order at the end).  The spatial tiles of a (co, ci) tile are split across blocks; every block stores its partial into
// its own slab of the workspace and wgrad_reduce_kernel adds the slabs in a fixed order: deterministic, no float atomics.
#include <stdlib.h>

#include "common.h"

namespace m355 {

struct Wgrad3Args {
  const half_t* dz; long dz_bs; int lddz;
  const half_t* x; long x_bs; int ldx;
  int H, W, Cin, Cout;
  float* out;            // dW [Cout][9][Cin] (splitk == 1) or the first slab
  long slab;             // floats between slabs
  int tiles_x, tiles_y, tiles_sp, tiles_per_split, ci_tiles, co_tiles;
  const half_t* zero;
};

namespace {

typedef short short4v __attribute__((ext_vector_type(4)));
typedef float float16v __attribute__((ext_vector_type(16)));

constexpr int TH = 8, TW = 16, PW = TW + 2;   // output tile, patch width; the patch has TH + 2 = 10 rows = 180 pixels

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

// 8 consecutive LDS rows (pixels) of one channel: rows +0..3 and +4..7
template <int ROWB>
__device__ __forceinline__ half8 tr_frag(const char* p) {
  const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)p);
  const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(p + 4 * ROWB));
  half8 r;
  const half4 l4 = *(const half4*)&lo, h4 = *(const half4*)&hi;
#pragma unroll
  for (int j = 0; j < 4; ++j) { r[j] = l4[j]; r[4 + j] = h4[j]; }
  return r;
}

template <int NCO, int NCI>
__global__ __launch_bounds__(256, 2) void conv_wgrad3_kernel(const Wgrad3Args a) {
  constexpr int ROWZ = 64 * NCO, ROWX = 64 * NCI;          // bytes per LDS row (pixel)
  constexpr int CPRZ = ROWZ / 16, CPRX = ROWX / 16;        // 16-byte chunks per row
  constexpr int RPIZ = 64 / CPRZ, RPIX = 64 / CPRX;        // rows per LDS-DMA instruction (1 KB)
  constexpr int ZI = TH * TW / RPIZ;                       // instructions per dZ tile
  constexpr int XI = (10 * PW + RPIX - 1) / RPIX;          // ... per patch (180 rows, rounded up)
  constexpr int ZBYTES = TH * TW * ROWZ, XBYTES = XI * 1024, STAGE = ZBYTES + XBYTES;
  constexpr int KS = 4 / (NCO * NCI);                      // waves that share one channel block: they split the tile rows
  constexpr int ZPW = (ZI + 3) / 4, XPW = (XI + 3) / 4;    // instructions per wave
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave % NCI, wm = (wave / NCI) % NCO, wk = wave / (NCI * NCO);

  int bid = blockIdx.x;
  const int per_split = a.co_tiles * a.ci_tiles;
  const int split = bid / per_split;
  bid -= split * per_split;
  const int ci_tile = bid / a.co_tiles, co_tile = bid - ci_tile * a.co_tiles;
  const int co_base = co_tile * 32 * NCO, ci_base = ci_tile * 32 * NCI;
  const int t0 = split * a.tiles_per_split;
  int t1 = t0 + a.tiles_per_split;
  if (t1 > a.tiles_sp) t1 = a.tiles_sp;
  if (t0 >= t1) return;

  // ---- loader: per instruction this lane owns LDS (row, slot); the source chunk is slot ^ f(row)
  const int zrr = lane / CPRZ, zslot = lane % CPRZ, xrr = lane / CPRX, xslot = lane % CPRX;
  int z_rel[ZPW], x_rel[XPW];           // element offset relative to the tile's first pixel / the patch's first pixel
  int z_ty[ZPW], z_tx[ZPW], x_py[XPW], x_px[XPW];
  bool z_ch_ok[ZPW], x_ch_ok[XPW];
#pragma unroll
  for (int i = 0; i < ZPW; ++i) {
    const int R = (wave + 4 * i) * RPIZ + zrr;
    const int ch = (ROWZ == 128) ? (zslot ^ (4 * ((R >> 1) & 1))) : zslot;
    z_ty[i] = R >> 4; z_tx[i] = R & 15;
    z_rel[i] = (z_ty[i] * a.W + z_tx[i]) * a.lddz + co_base + ch * 8;
    z_ch_ok[i] = co_base + ch * 8 < a.Cout;
  }
#pragma unroll
  for (int i = 0; i < XPW; ++i) {
    const int R = (wave + 4 * i) * RPIX + xrr;
    const int ch = (ROWX == 128) ? (xslot ^ (4 * ((R >> 1) & 1))) : xslot;
    x_py[i] = R / PW; x_px[i] = R - x_py[i] * PW;
    x_rel[i] = (x_py[i] * a.W + x_px[i]) * a.ldx + ci_base + ch * 8;
    x_ch_ok[i] = ci_base + ch * 8 < a.Cin && R < 10 * PW;
  }
  auto issue = [&](int t, int buf) __attribute__((always_inline)) {
    const int tx = t % a.tiles_x;
    const int rest = t / a.tiles_x;
    const int ty = rest % a.tiles_y, b = rest / a.tiles_y;
    const int y0 = ty * TH, x0 = tx * TW;
    char* zb = smem + buf * STAGE;
    char* xb = zb + ZBYTES;
    const half_t* zsrc = a.dz + (long)b * a.dz_bs + ((long)y0 * a.W + x0) * a.lddz;
    const half_t* xsrc = a.x + (long)b * a.x_bs + ((long)(y0 - 1) * a.W + (x0 - 1)) * a.ldx;
#pragma unroll
    for (int i = 0; i < ZPW; ++i) {
      if (wave + 4 * i >= ZI) continue;
      const bool ok = z_ch_ok[i] && y0 + z_ty[i] < a.H && x0 + z_tx[i] < a.W;
      glds16(ok ? zsrc + z_rel[i] : a.zero, zb + (wave + 4 * i) * 1024);
    }
#pragma unroll
    for (int i = 0; i < XPW; ++i) {
      if (wave + 4 * i >= XI) continue;
      const int yy = y0 - 1 + x_py[i], xx = x0 - 1 + x_px[i];
      const bool ok = x_ch_ok[i] && (unsigned)yy < (unsigned)a.H && (unsigned)xx < (unsigned)a.W;
      glds16(ok ? xsrc + x_rel[i] : a.zero, xb + (wave + 4 * i) * 1024);
    }
  };

  // ---- fragment read offsets.  16-lane group G = lane >> 4: columns 16 (G & 1) + li of the 32-channel block, pixels 8 (G >> 1) + ...
  const int G = lane >> 4, li = lane & 15, q = li >> 2, p = li & 3;
  const int kg = G >> 1, c16 = G & 1;
  // dZ: row = 16 c + 8 kg + q (+ 4): (row >> 1) & 1 = (q >> 1) & 1 whatever c and kg
  const int zchunk = wm * 4 + 2 * c16 + (p >> 1);
  const int zoff = (8 * kg + q) * ROWZ + ((ROWZ == 128 ? (zchunk ^ (4 * ((q >> 1) & 1))) : zchunk) << 4) + (p & 1) * 8;
  // patch: row = base + 8 kg + q with base = (c + kh) * 18 + kw: the swizzle bit is ((base + q) >> 1) & 1, one variant per base & 3
  const int xchunk = wn * 4 + 2 * c16 + (p >> 1);
  int xoff[4];
#pragma unroll
  for (int v = 0; v < 4; ++v)
    xoff[v] = (8 * kg + q) * ROWX + ((ROWX == 128 ? (xchunk ^ (4 * (((v + q) >> 1) & 1))) : xchunk) << 4) + (p & 1) * 8;

  float16v acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[t][j] = 0.f;

  issue(t0, 0);
  int buf = 0;
  for (int t = t0; t < t1; ++t) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();     // tile t has landed for every wave; every wave is done reading the other buffer
    if (t + 1 < t1) issue(t + 1, buf ^ 1);
    const char* zb = smem + buf * STAGE;
    const char* xb = zb + ZBYTES;
#pragma unroll
    for (int cc = 0; cc < TH / KS; ++cc) {
      const int c = cc * KS + wk;                       // tile row = K chunk of 16 pixels
      const half8 af = tr_frag<ROWZ>(zb + c * 16 * ROWZ + zoff);
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int base = (c + kh) * PW + kw;
          const half8 bf = tr_frag<ROWX>(xb + base * ROWX + xoff[base & 3]);
          acc[kh * 3 + kw] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bf, acc[kh * 3 + kw], 0, 0, 0);
        }
    }
    buf ^= 1;
  }

  // ---- rows split across waves: add the partials of waves wk = 1 .. KS - 1 into wave wk = 0, in wave order, through LDS
  if (KS > 1) {
    float* red = (float*)smem;      // [KS - 1][NCO * NCI][64 lanes][16]
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      __syncthreads();
      if (wk > 0) {
        float* dst = red + (((wk - 1) * (NCO * NCI) + wm * NCI + wn) * 64 + lane) * 16;
#pragma unroll
        for (int j = 0; j < 16; j += 4) *(float4v*)(dst + j) = float4v{acc[t][j], acc[t][j + 1], acc[t][j + 2], acc[t][j + 3]};
      }
      __syncthreads();
      if (wk == 0) {
#pragma unroll
        for (int s = 1; s < KS; ++s) {
          const float* src = red + (((s - 1) * (NCO * NCI) + wm * NCI + wn) * 64 + lane) * 16;
#pragma unroll
          for (int j = 0; j < 16; ++j) acc[t][j] += src[j];
        }
      }
    }
    if (wk > 0) return;
  }

  // ---- D[row = co][col = ci]: lane holds ci = lane & 31, co = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
  float* const outp = a.out + (long)split * a.slab;
  const int ci = ci_base + wn * 32 + (lane & 31);
  if (ci < a.Cin) {
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co_base + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (co < a.Cout) outp[((long)co * 9 + t) * a.Cin + ci] = acc[t][r];
      }
  }
}

struct Plan3 { int nco, nci, co_tiles, ci_tiles, tiles_x, tiles_y, tiles_sp, splitk, tps; };

Plan3 plan3(int B, int H, int W, int Cin, int Cout) {
  Plan3 p{};
  p.nco = Cout > 32 ? 2 : 1;
  p.nci = Cin > 32 ? 2 : 1;
  static const int shrink = getenv("M355_WGRAD3_SHRINK") ? atoi(getenv("M355_WGRAD3_SHRINK")) : 0;   // experiments: 1 halve ci, 2 halve co, 3 both when the layer is one tile
  if (Cout <= 64 && Cin <= 64) {
    if (shrink & 1) p.nci = 1;
    if (shrink & 2) p.nco = 1;
  }
  p.co_tiles = (Cout + 32 * p.nco - 1) / (32 * p.nco);
  p.ci_tiles = (Cin + 32 * p.nci - 1) / (32 * p.nci);
  p.tiles_x = (W + TW - 1) / TW;
  p.tiles_y = (H + TH - 1) / TH;
  p.tiles_sp = B * p.tiles_x * p.tiles_y;
  // Splits: about two blocks per CU, at least 8 spatial tiles per block (its prologue and its 9 x 32 x 32 x 4-byte-per-wave
  // store amortise), and no more slab bytes than ~96 MB per layer (every slab is written once and read once by the reduction).
  static const int target = getenv("M355_WGRAD3_BLOCKS") ? atoi(getenv("M355_WGRAD3_BLOCKS")) : 512;
  static const int min_tiles = getenv("M355_WGRAD3_MINTILES") ? atoi(getenv("M355_WGRAD3_MINTILES")) : 8;
  static const long slab_cap = (getenv("M355_WGRAD3_SLABMB") ? atol(getenv("M355_WGRAD3_SLABMB")) : 96) << 20;
  int sk = (target + p.co_tiles * p.ci_tiles - 1) / (p.co_tiles * p.ci_tiles);
  if (sk > p.tiles_sp / min_tiles) sk = p.tiles_sp / min_tiles;
  const long dw_bytes = (long)Cout * 9 * Cin * 4;
  if ((long)sk * dw_bytes > slab_cap) sk = (int)(slab_cap / dw_bytes);
  if (sk > p.tiles_sp) sk = p.tiles_sp;
  if (sk < 1) sk = 1;
  p.tps = (p.tiles_sp + sk - 1) / sk;
  p.splitk = (p.tiles_sp + p.tps - 1) / p.tps;
  return p;
}

template <int NCO, int NCI>
int launch3(const Wgrad3Args& a, int blocks, hipStream_t s) {
  constexpr int ROWZ = 64 * NCO, ROWX = 64 * NCI;
  constexpr int XI = (10 * PW + 64 / (ROWX / 16) - 1) / (64 / (ROWX / 16));
  constexpr int LDS = 2 * (TH * TW * ROWZ + XI * 1024);
  static_assert(LDS >= (4 / (NCO * NCI) - 1) * (NCO * NCI) * 4096, "the end-of-block reduction reuses the stages");
  if (LDS > 65536) {
    static bool set = false;
    if (!set) {
      hipError_t e = hipFuncSetAttribute((const void*)conv_wgrad3_kernel<NCO, NCI>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
      if (e != hipSuccess) return (int)e;
      set = true;
    }
  }
  hipLaunchKernelGGL((conv_wgrad3_kernel<NCO, NCI>), dim3(blocks), dim3(256), LDS, s, a);
  return (int)hipGetLastError();
}

}  // namespace

// Eligibility: 3x3 / stride 1 / pad 1, at least two tile columns (a 20-pixel-wide map would compute 32), 32-bit offsets.
bool conv_wgrad3_ok(int B, int H, int W, int Cin, int Cout, int ksize, int stride, int pad, int lddz, int ldx) {
  static const bool off = getenv("M355_NO_WGRAD3") != nullptr;
  if (off || ksize != 3 || stride != 1 || pad != 1 || Cin % 8 || Cout % 8 || lddz % 8 || ldx % 8) return false;
  if (W < 2 * TW || (long)H * W * (lddz > ldx ? lddz : ldx) >= (1L << 30)) return false;
  const int waste_num = ((W + TW - 1) / TW) * TW * (((H + TH - 1) / TH) * TH);
  return waste_num * 4 <= H * W * 5;   // at most 25 % of the tile area outside the map
}

size_t conv_wgrad3_workspace_bytes(int B, int H, int W, int Cin, int Cout) {
  const Plan3 p = plan3(B, H, W, Cin, Cout);
  return p.splitk > 1 ? (size_t)p.splitk * Cout * 9 * Cin * sizeof(float) : 0;
}

// Launches the patch kernel only (the caller adds the slabs); *splitk receives the number of slabs written.
int launch_conv_wgrad3(const half_t* dz, long dz_bs, int lddz, const half_t* x, long x_bs, int ldx, int B, int H, int W, int Cin,
                       int Cout, float* dw, const half_t* zero, float* ws, size_t ws_bytes, int* splitk, hipStream_t s) {
  const Plan3 p = plan3(B, H, W, Cin, Cout);
  if (p.splitk > 1 && (!ws || ws_bytes < (size_t)p.splitk * Cout * 9 * Cin * sizeof(float))) return -3;
  Wgrad3Args a{};
  a.dz = dz; a.dz_bs = dz_bs; a.lddz = lddz; a.x = x; a.x_bs = x_bs; a.ldx = ldx;
  a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
  a.out = p.splitk > 1 ? ws : dw;
  a.slab = (long)Cout * 9 * Cin;
  a.tiles_x = p.tiles_x; a.tiles_y = p.tiles_y; a.tiles_sp = p.tiles_sp; a.tiles_per_split = p.tps;
  a.ci_tiles = p.ci_tiles; a.co_tiles = p.co_tiles;
  a.zero = zero;
  *splitk = p.splitk;
  const int blocks = p.co_tiles * p.ci_tiles * p.splitk;
  if (p.nco == 2 && p.nci == 2) return launch3<2, 2>(a, blocks, s);
  if (p.nco == 2) return launch3<2, 1>(a, blocks, s);
  if (p.nci == 2) return launch3<1, 2>(a, blocks, s);
  return launch3<1, 1>(a, blocks, s);
}

}  // namespace m355
