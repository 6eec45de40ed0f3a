// Weight gradient of an NHWC fp16 convolution on CDNA4 MFMA (gfx950), fp32 accumulate / fp32 output.
// Replaces the wgrad half of aten conv2d's autograd reached from SegmentationTrainer's backward
// (call site BscanBased/yolo_seg_train.py:12).
//
//   dW[co][n] = sum_px dZ[px][co] * Xcol[px][n],      n = (kh*KS + kw)*Cin + ci      (KRSC layout, = the
//   packed forward-weight order), px = (b, ho, wo), Xcol[px][n] = X[b, ho*s - p + kh, wo*s - p + kw, ci].
//
// A GEMM whose reduction dimension is the PIXEL axis: both operands are row = pixel images in memory (NHWC),
// i.e. K-major the "wrong" way for MFMA fragments (a lane needs 8 consecutive pixels of ONE channel).  They
// are staged as they lie -- 64 pixel rows x 128 channels (256 B rows) by LDS-DMA with a per-lane gathered
// source address, zero page for padding / out-of-range -- and read back with gfx950's transposed LDS read
// ds_read_b64_tr_b16 (4 rows x 16 columns -> column-major, two reads = the 8 k-values of a fragment).  The
// 16-byte chunk index of a row is XOR-swizzled with f(row) = 2*(row&3) | ((row>>3)&1)<<3 (source side + read
// side), found by exhaustive search: every transposed read is bank-conflict free.
// Block = 128 co x 128 n output tile, 4 waves (64 x 64 each, 16 accumulator tiles), K step = 64 pixels, two
// LDS stages; the pixel axis is split across blocks (split-K).  DETERMINISTIC reduction: every split writes its partial
// tile with plain stores into its own slab of a caller-provided workspace ([splitk][Cout][N] fp32) and a second kernel
// adds the slabs in a fixed order (wgrad_reduce_kernel) -- bitwise reproducible gradients (float atomics made two runs of one batch differ, and
// a 60-layer fp16-storage backward amplifies such last-bit differences to 1e-2 in the early layers: DESIGN.md section 8).
// One split (small layers): the tile goes straight to dW.
#include <stdlib.h>

#include "common.h"

namespace m355 {

struct WgradArgs {
  const half_t* dz; long dz_bstride; int lddz;  // dZ (B,Ho,Wo,Cout) slice
  const half_t* x;  long x_bstride;  int ldx;   // X  (B,Hi,Wi,Cin) slice
  int Hi, Wi, Cin, Ho, Wo, Cout;
  int ksize, stride, pad;
  int M;        // B*Ho*Wo
  int N;        // ksize*ksize*Cin
  float* dw;    // [Cout][N] fp32 (written by the reduce kernel, or directly when splitk == 1)
  float* ws;    // [splitk][Cout][N] fp32 partial slabs (splitk > 1)
  int splitk, steps_per_split;
  const half_t* zero;
};

namespace {

typedef short short4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

__device__ __forceinline__ void fast_divmod(int n, int d, float inv_d, int& q, int& r) {
  q = (int)((float)n * inv_d);
  r = n - q * d;
  if (r < 0) { r += d; --q; }
  if (r >= d) { r -= d; ++q; }
}

__device__ __forceinline__ half8 tr_frag(const char* p) {
  // two transposed reads: rows +0..3 and +4..7 (4 rows = 1024 bytes apart) -> 8 k-values of one column
  const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)p);
  const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(p + 1024));
  half8 r;
  const half4 l4 = *(const half4*)&lo, h4 = *(const half4*)&hi;
#pragma unroll
  for (int j = 0; j < 4; ++j) { r[j] = l4[j]; r[4 + j] = h4[j]; }
  return r;
}

constexpr int KST = 64;            // pixels per K step
constexpr int ROWB2 = 256;         // LDS row = 128 channels
constexpr int OPB = KST * ROWB2;   // bytes per operand per stage (16 KB)
constexpr int STAGE2 = 2 * OPB;

template <int KS>
__global__ __launch_bounds__(256, 2) void conv_wgrad_kernel(const WgradArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int co_tiles = (a.Cout + 127) / 128, n_tiles = (a.N + 127) / 128;
  int bid = blockIdx.x;
  const int split = bid / (co_tiles * n_tiles);
  bid -= split * co_tiles * n_tiles;
  const int n_tile = bid / co_tiles, co_tile = bid - n_tile * co_tiles;
  const int co_base = co_tile * 128, n_base = n_tile * 128;
  const int steps_total = (a.M + KST - 1) / KST;
  const int step0 = split * a.steps_per_split;
  int nsteps = steps_total - step0;
  if (nsteps > a.steps_per_split) nsteps = a.steps_per_split;
  if (nsteps <= 0) return;

  // ---- loader: instr i of this wave fills rows R = 16*wave + 4*i + (lane>>4), slot = lane & 15;
  // source chunk = slot ^ f(R), f(R) = 2*(R&3) | ((R>>3)&1)<<3 = 2*(lane>>4) | ((i>>1)&1)<<3
  const int lr = lane >> 4, slot = lane & 15;
  const int c_lo = slot ^ (2 * lr), c_hi = c_lo ^ 8;
  const int HoWo = a.Ho * a.Wo;
  const float inv_howo = 1.0f / (float)HoWo, inv_wo = 1.0f / (float)a.Wo;
  // per chunk-variant: A channel offset and B (tap, ci)
  int a_off[2], b_kh[2], b_kw[2], b_ci[2];
  bool a_ok[2], b_ok[2];
#pragma unroll
  for (int v = 0; v < 2; ++v) {
    const int c = v ? c_hi : c_lo;
    a_off[v] = co_base + c * 8;
    a_ok[v] = a_off[v] < a.Cout;
    const int n = n_base + c * 8;
    b_ok[v] = n < a.N;
    const int tap = n / a.Cin;
    b_ci[v] = n - tap * a.Cin;
    b_kh[v] = tap / KS;
    b_kw[v] = tap - b_kh[v] * KS;
  }
  auto stage = [&](int t, int buf) {
    char* sa = smem + buf * STAGE2;
    char* sb = sa + OPB;
    const int k0 = (step0 + t) * KST;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int v = i >> 1;
      const int m = k0 + 16 * wave + 4 * i + lr;
      const bool mv = m < a.M;
      int b, pix, ho, wo;
      fast_divmod(mv ? m : 0, HoWo, inv_howo, b, pix);
      fast_divmod(pix, a.Wo, inv_wo, ho, wo);
      const half_t* srcA = (mv && a_ok[v]) ? (a.dz + (long)b * a.dz_bstride + (long)pix * a.lddz + a_off[v]) : a.zero;
      const int hi = ho * a.stride - a.pad + b_kh[v], wi = wo * a.stride - a.pad + b_kw[v];
      const bool okb = mv && b_ok[v] && (unsigned)hi < (unsigned)a.Hi && (unsigned)wi < (unsigned)a.Wi;
      const half_t* srcB = okb ? (a.x + (long)b * a.x_bstride + ((long)hi * a.Wi + wi) * a.ldx + b_ci[v]) : a.zero;
      glds16(srcA, sa + (16 * wave + 4 * i) * ROWB2);
      glds16(srcB, sb + (16 * wave + 4 * i) * ROWB2);
    }
  };

  // ---- fragment addresses: lane = (k-group G = lane>>4, li = lane&15 -> q = li>>2, p = li&3)
  const int wm = wave >> 1, wn = wave & 1;
  const int G = lane >> 4, li = lane & 15, q = li >> 2, p = li & 3;
  const int frow = 8 * G + q;                             // + 4 for the second read, + 32 for ks = 1
  const int fsw = (2 * q) | ((G & 1) << 3);               // f(row) for every row this lane addresses
  int aoff[4], boff[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int chA = (wm * 64 + t * 16 + 4 * p) >> 3, chB = (wn * 64 + t * 16 + 4 * p) >> 3;
    aoff[t] = frow * ROWB2 + ((chA ^ fsw) << 4) + (p & 1) * 8;
    boff[t] = OPB + frow * ROWB2 + ((chB ^ fsw) << 4) + (p & 1) * 8;
  }

  float4v acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = float4v{0.f, 0.f, 0.f, 0.f};

  stage(0, 0);
  for (int t = 0; t < nsteps; ++t) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t + 1 < nsteps) stage(t + 1, (t + 1) & 1);
    const char* sb = smem + (t & 1) * STAGE2;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      half8 af[4], bf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = tr_frag(sb + aoff[i] + ks * 32 * ROWB2);
#pragma unroll
      for (int i = 0; i < 4; ++i) bf[i] = tr_frag(sb + boff[i] + ks * 32 * ROWB2);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
  }

  // ---- this split's partial tile: D[row = co 4g+j][col = n l15], plain stores into the split's slab (or dW itself)
  const int l15 = lane & 15, g = lane >> 4;
  float* const outp = a.splitk > 1 ? a.ws + (long)split * a.Cout * a.N : a.dw;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n_base + wn * 64 + j * 16 + l15;
      if (n >= a.N) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = co_base + wm * 64 + i * 16 + g * 4 + r;
        if (co < a.Cout) outp[(long)co * a.N + n] = acc[i][j][r];
      }
    }
}

// dW[i] = sum over the splits of ws[s][i] in a FIXED association (bitwise reproducible): 16 lanes x 16 split groups per block;
// group g adds slabs g, g + 16, g + 32, ... in that order, the 16 group sums are added pairwise in LDS ((0+1)+(2+3))+...
// One thread per element walking every slab (the first version) ran 9 - 36 blocks for the small layers: 123 us to add 512
// slabs of 36 KB, twice the time of the gradient kernel itself.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* ws, float* dw, long n, int splitk) {
  __shared__ float4v part[16][16];
  const int tx = threadIdx.x & 15, g = threadIdx.x >> 4;
  const long i = ((long)blockIdx.x * 16 + tx) * 4;
  float4v acc = {0.f, 0.f, 0.f, 0.f};
  if (i + 4 <= n) {
    for (int s = g; s < splitk; s += 16) {
      const float4v v = *(const float4v*)(ws + (long)s * n + i);
      acc[0] += v[0]; acc[1] += v[1]; acc[2] += v[2]; acc[3] += v[3];
    }
  } else if (i < n) {
    for (int s = g; s < splitk; s += 16)
      for (int j = 0; j < 4 && i + j < n; ++j) acc[j] += ws[(long)s * n + i + j];
  }
  part[g][tx] = acc;
  __syncthreads();
#pragma unroll
  for (int w = 8; w >= 1; w >>= 1) {
    if (g < w) {
      const float4v o = part[g + w][tx];
      float4v m = part[g][tx];
      m[0] += o[0]; m[1] += o[1]; m[2] += o[2]; m[3] += o[3];
      part[g][tx] = m;
    }
    __syncthreads();
  }
  if (g == 0 && i < n) {
    const float4v r = part[0][tx];
    if (i + 4 <= n) *(float4v*)(dw + i) = r;
    else for (int j = 0; j < 4 && i + j < n; ++j) dw[i + j] = r[j];
  }
}

// ---------------------------------------------------------------------------------------------------------
// Weight gradient of the STEM (3x3 / s2 / p1 on the 8-channel padded input rows, Cout <= 64; upstream model.0 reached from
// yolo_seg_train.py:12's backward).  It is the LAST kernel of a training step -- dZ of layer 0 exists only when everything else is
// done -- so nothing overlaps it, and the pixel-axis GEMM above gives it a 128 x 128 tile for a 32 x 72 result: 485 us at batch 64
// @640 for 840 MB of operands (1.7 TB/s), the exposed tail of the step.
// Here a WAVE owns a stream of chunks (64 consecutive output pixels of one row): dZ [64 px][Cout] and the three input rows
// [3][129 px][8 ch] go to a wave-private LDS image through registers (the next chunk's loads are in flight while this one is
// multiplied), the fragments are gathered with 2-byte LDS reads (the matrices are tiny: 12 MFMAs per chunk), accumulators live in
// registers for the whole stream.  D[co][n], n = (tap, ci) = the KRSC column.  The four waves of a block add their tiles in wave
// order and the block writes ONE partial slab; wgrad_reduce_kernel adds the slabs in block order: bitwise reproducible.
// ---------------------------------------------------------------------------------------------------------
typedef float float16v __attribute__((ext_vector_type(16)));
constexpr int SW_PX = 64;                  // output pixels per chunk
constexpr int SW_XCOLS = 2 * SW_PX + 1;    // input columns a chunk touches (2 wo0 - 1 .. 2 wo0 + 127)
constexpr int SW_XPITCH = SW_XCOLS + 1;    // LDS pitch of an input row in pixels (16 bytes each)

template <int MB>
__global__ __launch_bounds__(256) void wgrad_stem_kernel(const half_t* dz, long dz_bs, int lddz, const half_t* x, long x_bs, int B, int Ho,
                                                         int Wo, int Cout, float* out, int nsplit) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int ZROW = MB * 64;                                   // bytes per pixel row of the dZ image (MB * 32 channels)
  constexpr int WBYTES = SW_PX * ZROW + 3 * SW_XPITCH * 16;       // one wave's images
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, kg = lane >> 5;
  char* const zb = smem + wave * WBYTES;
  char* const xb = zb + SW_PX * ZROW;
  const int Hi = 2 * Ho, Wi = 2 * Wo;
  const int segs = Wo / SW_PX;
  const long total = (long)B * Ho * segs;
  const long nwv = (long)gridDim.x * 4;
  const int cpp = Cout / 8;                                       // 16-byte chunks per dZ pixel
  constexpr int ZL = MB * 4;                                      // dZ loads per lane and chunk at the full width
  // channels Cout .. MB * 32 of the dZ image stay zero (Cout = 16, 48): cleared once
  for (int i = lane; i < SW_PX * ZROW / 16; i += 64) *(half8*)(zb + i * 16) = half8{0, 0, 0, 0, 0, 0, 0, 0};

  half8 zr[ZL], xr[9];
  auto fetch = [&](long c) __attribute__((always_inline)) {
    const int seg = (int)(c % segs);
    const long r = c / segs;
    const int ho = (int)(r % Ho);
    const long b = r / Ho;
    const int wo0 = seg * SW_PX;
    const half_t* zrow = dz + b * dz_bs + ((long)ho * Wo + wo0) * lddz;
#pragma unroll
    for (int i = 0; i < ZL; ++i) {
      const int q = i * 64 + lane;
      const int px = q / cpp, cc = q - px * cpp;
      zr[i] = half8{0, 0, 0, 0, 0, 0, 0, 0};
      if (px < SW_PX) zr[i] = *(const half8*)(zrow + (long)px * lddz + cc * 8);
    }
    const half_t* xim = x + b * x_bs;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int iy = 2 * ho - 1 + kh;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int t = j * 64 + lane, ix = 2 * wo0 - 1 + t;
        xr[kh * 3 + j] = half8{0, 0, 0, 0, 0, 0, 0, 0};
        if (t < SW_XCOLS && (unsigned)iy < (unsigned)Hi && (unsigned)ix < (unsigned)Wi) xr[kh * 3 + j] = *(const half8*)(xim + ((long)iy * Wi + ix) * 8);
      }
    }
  };
  auto stash = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < ZL; ++i) {
      const int q = i * 64 + lane;
      const int px = q / cpp, cc = q - px * cpp;
      if (px < SW_PX) *(half8*)(zb + px * ZROW + cc * 16) = zr[i];
    }
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int t = j * 64 + lane;
        if (t < SW_XCOLS) *(half8*)(xb + (kh * SW_XPITCH + t) * 16) = xr[kh * 3 + j];
      }
  };

  float16v acc[MB][3];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mb][g][i] = 0.f;
  // B fragment columns of this lane: n = 32 g + l31 -> tap = n / 8 (valid < 9), channel n % 8
  int boff[3];
  bool bval[3];
#pragma unroll
  for (int g = 0; g < 3; ++g) {
    const int n = 32 * g + l31, tap = n >> 3, ci = n & 7;
    bval[g] = tap < 9;
    const int kh = bval[g] ? tap / 3 : 0, kw = bval[g] ? tap - 3 * kh : 0;
    boff[g] = ((kh * SW_XPITCH + kw) * 8 + ci) * 2;                // + 32 bytes per output pixel (two input columns)
  }

  long c = (long)blockIdx.x * 4 + wave;
  if (c < total) fetch(c);
  for (; c < total; c += nwv) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             // (this wave's reads of the previous chunk's images are done)
    stash();
    if (c + nwv < total) fetch(c + nwv);
#pragma unroll
    for (int ks = 0; ks < SW_PX / 16; ++ks) {
      const int p0 = 16 * ks + 8 * kg;                             // this lane's eight pixels (the K values of its fragments)
      half8 af[MB], bf[3];
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int j = 0; j < 8; ++j) af[mb][j] = *(const half_t*)(zb + (p0 + j) * ZROW + (mb * 32 + l31) * 2);
#pragma unroll
      for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int j = 0; j < 8; ++j) bf[g][j] = bval[g] ? *(const half_t*)(xb + boff[g] + (p0 + j) * 32) : (half_t)0.f;
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int g = 0; g < 3; ++g) acc[mb][g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[mb], bf[g], acc[mb][g], 0, 0, 0);
    }
  }
  // ---- the block's tile = its four waves' tiles added in wave order (through LDS, reusing the image memory)
  __syncthreads();
  float* const red = (float*)smem;                                 // [wave][MB * 32][96]
  constexpr int TILE = MB * 32 * 96;
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = mb * 32 + 8 * (i >> 2) + 4 * kg + (i & 3);
        red[wave * TILE + row * 96 + 32 * g + l31] = acc[mb][g][i];
      }
  __syncthreads();
  float* const slab = out + (nsplit > 1 ? (long)blockIdx.x * Cout * 72 : 0);
  for (int e = tid; e < Cout * 72; e += 256) {
    const int co = e / 72, n = e - co * 72;
    const float* r = red + co * 96 + n;
    slab[e] = ((r[0] + r[TILE]) + r[2 * TILE]) + r[3 * TILE];
  }
}

bool wgrad_stem_ok(int Hi, int Wi, int Cin, int Ho, int Wo, int Cout, int ksize, int stride, int pad, int lddz, int ldx, long x_bstride) {
  static const bool off = getenv("M355_NO_WGRAD_STEM") != nullptr;
  return !off && ksize == 3 && stride == 2 && pad == 1 && Cin == 8 && ldx == 8 && x_bstride == (long)Hi * Wi * 8 && Cout >= 8 && Cout <= 64 &&
         Cout % 8 == 0 && lddz >= Cout && Hi == 2 * Ho && Wi == 2 * Wo && Wo % SW_PX == 0;
}

// ---------------------------------------------------------------------------------------------------------
// Weight gradient of model.1 of the s scale (3x3 / s2 / p1, 32 -> 64 channels on the 320 x 320 map).  With its input gradient on
// conv_dgrad_s2c32.hip this launch became the critical path at the end of a training step: the pixel-axis GEMM gives it three
// 128 x 128 tiles for a 64 x 288 result (half of every tile is channel padding, the input patches are gathered once per tile):
// 418 us alone, 712 us beside the batch-norm backward of layer 0.
// Block-cooperative chunk stream: a chunk is 64 consecutive output pixels of one row; the block stages dZ [64 px][64 ch] and the three
// input rows [3][129 px][32 ch] in LDS through registers (the next chunk's loads are in flight during the multiply).  The 9 taps x 2
// output-channel blocks = 18 accumulator tiles (32 co x 32 ci each: the KRSC column block of tap t is n = 32 t .. 32 t + 31) are
// dealt to the four waves 5 / 5 / 4 / 4, so no cross-wave reduction exists: every wave gathers its A fragments (dZ transposed: eight
// 2-byte LDS reads) and its taps' B fragments and owns its tiles for the whole stream.  One partial slab per block,
// wgrad_reduce_kernel adds the slabs in block order: bitwise reproducible.
// ---------------------------------------------------------------------------------------------------------
constexpr int W2_PX = 64, W2_XCOLS = 2 * W2_PX + 1, W2_XPITCH = W2_XCOLS + 1;
constexpr int W2_ZBYTES = W2_PX * 128, W2_XBYTES = 3 * W2_XPITCH * 64, W2_LDS = W2_ZBYTES + W2_XBYTES;

// units (tap, channel block) of wave W: taps T0 .. T0 + NT - 1; the first / last tap may hold only one channel block
template <int W> struct W2Units;
template <> struct W2Units<0> { static constexpr int T0 = 0, NT = 3; static constexpr int mask[3] = {3, 3, 1}; };
template <> struct W2Units<1> { static constexpr int T0 = 2, NT = 3; static constexpr int mask[3] = {2, 3, 3}; };
template <> struct W2Units<2> { static constexpr int T0 = 5, NT = 2; static constexpr int mask[3] = {3, 3, 0}; };
template <> struct W2Units<3> { static constexpr int T0 = 7, NT = 2; static constexpr int mask[3] = {3, 3, 0}; };

template <int W>
__device__ __forceinline__ void wgrad_s2c32_run(const half_t* dz, long dz_bs, int lddz, const half_t* x, long x_bs, int ldx, int B, int Ho, int Wo,
                                                float* slab, char* smem) {
  using U = W2Units<W>;
  const int tid = threadIdx.x, lane = tid & 63;
  const int l31 = lane & 31, kg = lane >> 5;
  char* const zb = smem;
  char* const xb = smem + W2_ZBYTES;
  const int Hi = 2 * Ho, Wi = 2 * Wo;
  const int segs = (Wo + W2_PX - 1) / W2_PX;                   // (the last chunk of a row may be partial: its missing dZ pixels are zeros)
  const long total = (long)B * Ho * segs;
  half8 zr[2], xr[7];
  auto fetch = [&](long c) __attribute__((always_inline)) {
    const int seg = (int)(c % segs);
    const long r = c / segs;
    const int ho = (int)(r % Ho);
    const long b = r / Ho;
    const int wo0 = seg * W2_PX;
    const half_t* zrow = dz + b * dz_bs + ((long)ho * Wo + wo0) * lddz;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int q = tid + 256 * u;
      zr[u] = half8{0, 0, 0, 0, 0, 0, 0, 0};
      if (wo0 + (q >> 3) < Wo) zr[u] = *(const half8*)(zrow + (long)(q >> 3) * lddz + (q & 7) * 8);
    }
    const half_t* xim = x + b * x_bs;
#pragma unroll
    for (int u = 0; u < 7; ++u) {
      const int q = tid + 256 * u;
      const int row = q / (W2_XCOLS * 4), rem = q - row * (W2_XCOLS * 4);
      const int px = rem >> 2, cg = rem & 3;
      const int iy = 2 * ho - 1 + row, ix = 2 * wo0 - 1 + px;
      xr[u] = half8{0, 0, 0, 0, 0, 0, 0, 0};
      if (row < 3 && (unsigned)iy < (unsigned)Hi && (unsigned)ix < (unsigned)Wi) xr[u] = *(const half8*)(xim + ((long)iy * Wi + ix) * ldx + cg * 8);
    }
  };
  auto stash = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int q = tid + 256 * u;
      *(half8*)(zb + (q >> 3) * 128 + (q & 7) * 16) = zr[u];
    }
#pragma unroll
    for (int u = 0; u < 7; ++u) {
      const int q = tid + 256 * u;
      const int row = q / (W2_XCOLS * 4), rem = q - row * (W2_XCOLS * 4);
      if (row < 3) *(half8*)(xb + ((row * W2_XPITCH + (rem >> 2)) * 32 + (rem & 3) * 8) * 2) = xr[u];
    }
  };
  float16v acc[U::NT][2];
#pragma unroll
  for (int t = 0; t < U::NT; ++t)
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[t][mb][e] = 0.f;
  int boff[U::NT];
#pragma unroll
  for (int t = 0; t < U::NT; ++t) {
    const int tap = U::T0 + t, kh = tap / 3, kw = tap - 3 * kh;
    boff[t] = ((kh * W2_XPITCH + kw) * 32 + l31) * 2;          // + 128 bytes per output pixel (two input columns of 64 bytes)
  }
  constexpr int need = U::mask[0] | U::mask[1] | (U::NT > 2 ? U::mask[2] : 0);   // channel blocks this wave multiplies at all

  long c = blockIdx.x;
  if (c < total) fetch(c);
  for (; c < total; c += gridDim.x) {
    __syncthreads();                                           // every wave is done with the previous chunk's images
    stash();
    __syncthreads();
    if (c + gridDim.x < total) fetch(c + gridDim.x);
#pragma unroll
    for (int ks = 0; ks < W2_PX / 16; ++ks) {
      const int p0 = 16 * ks + 8 * kg;
      half8 af[2], bf[U::NT];
#pragma unroll
      for (int mb = 0; mb < 2; ++mb)
        if ((need >> mb) & 1) {
#pragma unroll
          for (int j = 0; j < 8; ++j) af[mb][j] = *(const half_t*)(zb + (p0 + j) * 128 + (mb * 32 + l31) * 2);
        }
#pragma unroll
      for (int t = 0; t < U::NT; ++t)
#pragma unroll
        for (int j = 0; j < 8; ++j) bf[t][j] = *(const half_t*)(xb + boff[t] + (p0 + j) * 128);
#pragma unroll
      for (int t = 0; t < U::NT; ++t)
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
          if ((U::mask[t] >> mb) & 1) acc[t][mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[mb], bf[t], acc[t][mb], 0, 0, 0);
    }
  }
#pragma unroll
  for (int t = 0; t < U::NT; ++t)
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
      if ((U::mask[t] >> mb) & 1) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int co = mb * 32 + 8 * (e >> 2) + 4 * kg + (e & 3);
          slab[co * 288 + (U::T0 + t) * 32 + l31] = acc[t][mb][e];
        }
      }
}

__global__ __launch_bounds__(256) void wgrad_s2c32_kernel(const half_t* dz, long dz_bs, int lddz, const half_t* x, long x_bs, int ldx, int B, int Ho,
                                                          int Wo, float* out, int nsplit) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float* const slab = out + (nsplit > 1 ? (long)blockIdx.x * 64 * 288 : 0);
  switch (wave) {
    case 0: wgrad_s2c32_run<0>(dz, dz_bs, lddz, x, x_bs, ldx, B, Ho, Wo, slab, smem); break;
    case 1: wgrad_s2c32_run<1>(dz, dz_bs, lddz, x, x_bs, ldx, B, Ho, Wo, slab, smem); break;
    case 2: wgrad_s2c32_run<2>(dz, dz_bs, lddz, x, x_bs, ldx, B, Ho, Wo, slab, smem); break;
    default: wgrad_s2c32_run<3>(dz, dz_bs, lddz, x, x_bs, ldx, B, Ho, Wo, slab, smem); break;
  }
}

constexpr int W2_SLABS = 512;   // partial slabs (= blocks) the workspace is sized for
bool wgrad_s2c32_ok(int Hi, int Wi, int Cin, int Ho, int Wo, int Cout, int ksize, int stride, int pad, int lddz, int ldx) {
  static const bool off = getenv("M355_NO_WGRAD_S2C32") != nullptr;
  return !off && ksize == 3 && stride == 2 && pad == 1 && Cin == 32 && Cout == 64 && ldx >= 32 && lddz >= 64 && Hi == 2 * Ho && Wi == 2 * Wo &&
         Wo >= W2_PX;
}

void wgrad_plan(int M, int Cout, int N, int* splitk, int* steps_per_split) {
  const int tiles = ((Cout + 127) / 128) * ((N + 127) / 128);
  const int steps_total = (M + KST - 1) / KST;
  // every block writes one 64 KB partial tile: the slab traffic of a layer is (blocks x 64 KB), so no more blocks than
  // fill three quarters of the chip once (256 CUs x 2 resident blocks); M355_WGRAD_BLOCKS overrides for tuning
  static const int target = getenv("M355_WGRAD_BLOCKS") ? atoi(getenv("M355_WGRAD_BLOCKS")) : 384;   // measured (s-seg b64 @640 step): 256 -> 49.0 ms, 384 -> 46.1, 512 -> 47.7, 1024 -> 49.3
  int sk = (target + tiles - 1) / tiles;
  if (sk > steps_total) sk = steps_total;
  if (sk < 1) sk = 1;
  *steps_per_split = (steps_total + sk - 1) / sk;
  *splitk = (steps_total + *steps_per_split - 1) / *steps_per_split;
}

}  // namespace

size_t conv_wgrad_workspace_bytes(int B, int Ho, int Wo, int Cin, int Cout, int ksize) {
  int sk = 1, sps = 1;
  const int N = ksize * ksize * Cin;
  wgrad_plan(B * Ho * Wo, Cout, N, &sk, &sps);
  size_t need = sk > 1 ? (size_t)sk * Cout * N * sizeof(float) : 0;
  if (ksize == 3 && Cin == 32 && Cout == 64 && Wo >= W2_PX) {   // (stride unknown here: cover wgrad_s2c32_kernel's slabs)
    const size_t n2 = (size_t)W2_SLABS * 64 * 288 * sizeof(float);
    if (n2 > need) need = n2;
  }
  if (ksize == 3 && conv_wgrad3_ok(B, Ho, Wo, Cin, Cout, 3, 1, 1, 8, 8)) {   // (stride unknown here: cover the stride-1 patch kernel too)
    const size_t n3 = conv_wgrad3_workspace_bytes(B, Ho, Wo, Cin, Cout);
    if (n3 > need) need = n3;
  }
  return need;
}

int launch_conv_wgrad(const half_t* dz, long dz_bstride, int lddz, const half_t* x, long x_bstride, int ldx, int B,
                      int Hi, int Wi, int Cin, int Ho, int Wo, int Cout, int ksize, int stride, int pad, float* dw,
                      const half_t* zero, float* ws, size_t ws_bytes, hipStream_t s) {
  if (ksize < 1 || ksize > 3 || Cin % 8 || Cout % 8 || lddz % 8 || ldx % 8) return -1;
  if (conv_wgrad3_ok(B, Hi, Wi, Cin, Cout, ksize, stride, pad, lddz, ldx) && Ho == Hi && Wo == Wi) {
    int sk = 1;
    const int rc = launch_conv_wgrad3(dz, dz_bstride, lddz, x, x_bstride, ldx, B, Hi, Wi, Cin, Cout, dw, zero, ws, ws_bytes, &sk, s);
    if (rc != 0) return rc;
    if (sk > 1) {
      const long n = (long)Cout * 9 * Cin;
      hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((int)((n + 63) / 64)), dim3(256), 0, s, ws, dw, n, sk);
    }
    return (int)hipGetLastError();
  }
  if (wgrad_s2c32_ok(Hi, Wi, Cin, Ho, Wo, Cout, ksize, stride, pad, lddz, ldx) && ws && ws_bytes >= (size_t)64 * 64 * 288 * sizeof(float)) {
    static int cus = 0;
    if (!cus) {
      int dev = 0;
      if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return -2;
    }
    const long chunks = (long)B * Ho * ((Wo + W2_PX - 1) / W2_PX);
    static const int per_cu_x2 = getenv("M355_W2_BLOCKS_X2") ? atoi(getenv("M355_W2_BLOCKS_X2")) : 4;   // blocks per CU x 2 (tuning)
    long nb = (long)per_cu_x2 * cus / 2;                     // two blocks per CU (registers), one slab each
    if (nb > W2_SLABS) nb = W2_SLABS;
    if (nb > chunks) nb = chunks;
    const long fit = (long)(ws_bytes / ((size_t)64 * 288 * sizeof(float)));
    if (nb > fit) nb = fit;
    hipLaunchKernelGGL(wgrad_s2c32_kernel, dim3((unsigned)nb), dim3(256), W2_LDS, s, dz, dz_bstride, lddz, x, x_bstride, ldx, B, Ho, Wo,
                       nb > 1 ? ws : dw, (int)nb);
    if (nb > 1) {
      const long n = 64L * 288;
      hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((int)((n + 63) / 64)), dim3(256), 0, s, ws, dw, n, (int)nb);
    }
    return (int)hipGetLastError();
  }
  if (wgrad_stem_ok(Hi, Wi, Cin, Ho, Wo, Cout, ksize, stride, pad, lddz, ldx, x_bstride)) {
    // the stem: one partial slab per block, as many blocks as the pixel-axis GEMM would use splits (the caller's workspace is sized for those)
    int sk = 1, sps = 1;
    wgrad_plan(B * Ho * Wo, Cout, 72, &sk, &sps);
    const long chunks = (long)B * Ho * (Wo / SW_PX);
    if ((long)sk * 4 > chunks) sk = (int)((chunks + 3) / 4);
    if (sk < 1) sk = 1;
    if (sk > 1 && (!ws || ws_bytes < (size_t)sk * Cout * 72 * sizeof(float))) return -3;
    const int mb = Cout > 32 ? 2 : 1;
    const int wbytes = SW_PX * mb * 64 + 3 * SW_XPITCH * 16;
    int lds = 4 * wbytes;
    if (lds < 4 * mb * 32 * 96 * 4) lds = 4 * mb * 32 * 96 * 4;
    auto k = mb == 2 ? wgrad_stem_kernel<2> : wgrad_stem_kernel<1>;
    if (lds > 65536) {
      hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(k, dim3(sk), dim3(256), lds, s, dz, dz_bstride, lddz, x, x_bstride, B, Ho, Wo, Cout, sk > 1 ? ws : dw, sk);
    if (sk > 1) {
      const long n = (long)Cout * 72;
      hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((int)((n + 63) / 64)), dim3(256), 0, s, ws, dw, n, sk);
    }
    return (int)hipGetLastError();
  }
  WgradArgs a{};
  a.dz = dz; a.dz_bstride = dz_bstride; a.lddz = lddz; a.x = x; a.x_bstride = x_bstride; a.ldx = ldx;
  a.Hi = Hi; a.Wi = Wi; a.Cin = Cin; a.Ho = Ho; a.Wo = Wo; a.Cout = Cout;
  a.ksize = ksize; a.stride = stride; a.pad = pad;
  a.M = B * Ho * Wo;
  if ((long)B * Ho * Wo >= (1 << 24)) return -1;
  a.N = ksize * ksize * Cin;
  a.dw = dw; a.zero = zero;
  const int tiles = ((Cout + 127) / 128) * ((a.N + 127) / 128);
  wgrad_plan(a.M, Cout, a.N, &a.splitk, &a.steps_per_split);
  if (a.splitk > 1 && (!ws || ws_bytes < (size_t)a.splitk * Cout * a.N * sizeof(float))) return -3;   // workspace too small
  a.ws = ws;
  const dim3 grid(tiles * a.splitk), block(256);
  const int lds = 2 * STAGE2;
  if (ksize == 1) {
    hipLaunchKernelGGL(conv_wgrad_kernel<1>, grid, block, lds, s, a);
  } else if (ksize == 2) {
    hipLaunchKernelGGL(conv_wgrad_kernel<2>, grid, block, lds, s, a);
  } else {
    hipLaunchKernelGGL(conv_wgrad_kernel<3>, grid, block, lds, s, a);
  }
  if (a.splitk > 1) {
    const long n = (long)Cout * a.N;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((int)((n + 63) / 64)), dim3(256), 0, s, ws, dw, n, a.splitk);
  }
  return (int)hipGetLastError();
}

}  // namespace m355
