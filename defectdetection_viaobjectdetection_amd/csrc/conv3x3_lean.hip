// 3x3 / stride-1 / pad-1 NHWC fp16 convolution, LEAN halo tiles: the structure of conv3x3_wide.hip (K depth 32 per
// step, rows of three steps with compile-time kw, weight ring slot = kw, affine patch pieces through a zero page,
// persistent tile walk with the epilogue of tile n under the prologue DMA of tile n+1) as a template over the
// channel block and the tile height, for the layers the 128 ch x 16x16 px tile does not fit:
//   <128, 8>  128 ch x  8x16 px, waves 2 (ch) x 2 (rows), 64 ch x 4 rows each (MT=4, NT=4); 51 KiB LDS, three blocks / CU
//             -> the 40x40 maps (16-row tiles would waste 44 % of the pixels there)
//   < 64,16>   64 ch x 16x16 px, waves 1 x 4, 64 ch x 4 rows each (MT=4, NT=4); 58.5 KiB LDS, two blocks / CU
//             -> the 64-channel layers of the 80x80 maps
// Measured (MI355X, batch 32, tools/op_table.py, tools/stamps_halo.py 22): EQUAL to conv3x3_halo.hip's K-64 kernels on
// every such layer (27-30 us each), so those stay the default and this template is opt-in (M355_LEAN=1, tile ids
// 21-24 in the test entry).  Why it is not faster: with 16 MFMAs per wave and step the loop has 97 instructions per
// step = 6.1 per MFMA, and a SIMD issues one instruction per four cycles whatever the number of resident waves
// (main loop 28.1 k cycles for 36 steps = 780 = 2 waves x 97.5 x 4): MFMA share 4 / 6.1 = 66 %, exactly what the
// stamps show for this kernel AND the K-64 one.  conv3x3_wide.hip sits at 3.8 instructions per MFMA and is MFMA-bound.
// A faster small-tile kernel has to get under four instructions per MFMA; none of (tile, K depth, waves) does that.
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

__device__ __forceinline__ float silu_f(float v) { return m355_silu(v); }

// s_waitcnt immediate (gfx9 encoding): vmcnt(n) lgkmcnt(0), expcnt untouched.  The builtin (unlike inline asm) is
// visible to the compiler's own wait-count insertion, which then does not re-wait for LDS reads issued before it.
#define WAITCNT_VM_LGKM0(n) ((((n) & 0xf) | (((n) >> 4) << 14) | (7 << 4)))

constexpr int TS = 16;                // output tile width
constexpr int PP = 20;                // patch row pitch in pixels (18 used)
constexpr int ROWB = 64;              // LDS row = 32 halves
constexpr int NWB = 3;                // weight ring: the slot of a step is its kw
constexpr int MT = 4;

template <int BCH, int TH>
struct Lean {
  static constexpr int PH = TH + 2;
  static constexpr int NTP = PH / 2;                     // taps that carry a patch piece (two patch rows each)
  static constexpr int PROWS = 40 * (NTP - 1) + 48;      // + the rows the last piece spills into (never read)
  static constexpr int PATCH_BYTES = PROWS * ROWB;
  static constexpr int WBUF = BCH * ROWB;
  static constexpr int W_IT = BCH / 64;                  // weight pieces per wave per step
  static constexpr int WCH = BCH / 64, WPX = 4 / WCH;    // wave grid: channels x image rows
  static constexpr int NT = TH / WPX;                    // image rows per wave
  static constexpr int LDS_BYTES = 2 * PATCH_BYTES + NWB * WBUF + 2 * BCH * 4;
  static constexpr int BLOCKS = LDS_BYTES * 3 <= 160 * 1024 ? 3 : 2;
};

template <int BCH, int TH>
__global__ __launch_bounds__(256, (Lean<BCH, TH>::BLOCKS)) void conv3x3_lean_kernel(const ConvArgs a, int tiles_x, int tiles_y,
                                                                                   int nchunks, int ntiles) {
  using L = Lean<BCH, TH>;
  constexpr int PH = L::PH, NTP = L::NTP, PATCH_BYTES = L::PATCH_BYTES, WBUF = L::WBUF, W_IT = L::W_IT, NT = L::NT;
  constexpr int WPX = L::WPX, HALF = NT / 2;
  static_assert(NT >= 4 && NT % 2 == 0, "the pinned schedule needs 1 + 4 + NT/2 MFMAs in the second half");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const wbase = smem + 2 * PATCH_BYTES;
  float* const sbias = (float*)(wbase + NWB * WBUF);   // two buffers of BCH floats (tile parity)

  unsigned long long st0 = 0, st1 = 0, st2 = 0, rt0 = 0, sa = 0, sb = 0, sc = 0, sd = 0, sa2 = 0;
  if (a.stamps) {
    st0 = __builtin_amdgcn_s_memtime();
    rt0 = __builtin_amdgcn_s_memrealtime();
  }
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lrow = lane >> 2;                         // row inside a 16-row DMA piece
  const int lslot = lane & 3;                         // 16-byte slot inside the row
  const int H = a.Hi, W = a.Wi;
  const int tiles_ch = (a.Cout + BCH - 1) / BCH;

  // ---- fragment addressing (tile independent)
  const int wch = wave / WPX, wpx = wave % WPX;
  const int l15 = lane & 15, g = lane >> 4;
  const int aoff = (wch * 64 + l15) * ROWB + ((g ^ (((l15 >> 2) & 1) << 1)) << 4);   // + mt * 1024 (immediate)
  const int pb = (wpx * NT) * PP + l15;                                                // patch row of (image row 0, x = l15)
  const int g16 = g << 4;

  // ---- patch streaming lane constants (waves 0-2): LDS row 40 t + r0 at tap t
  const int r0 = wave * 16 + lrow;
  const int pdy = r0 / PP, ppx = r0 - pdy * PP;
  const int pcc = lslot ^ (((r0 >> 2) & 1) << 1);                 // (40 t + r0) >> 2 has the parity of r0 >> 2
  // ---- weight streaming lane constant: LDS row R = i*64 + wave*16 + lrow holds permuted channel chl of block i
  int wlane;   // byte offset of this lane's 16 bytes inside the weight matrix of a channel tile (piece i adds 64 rows)
  {
    const int Rl = wave * 16 + lrow;
    const int mt = Rl >> 4, r = Rl & 15;
    const int chl = (mt >> 1) * 32 + (r >> 2) * 8 + (mt & 1) * 4 + (r & 3);
    const int cc = lslot ^ (((Rl >> 2) & 1) << 1);
    wlane = (chl * a.Kpad + cc * 8) * 2;
  }
  const long wblk = (long)64 * a.Kpad * 2;                        // bytes between the two 64-channel blocks
  const int prow_bytes = 2 * W * a.ldx * 2;                       // two image rows, in bytes

  // ---- persistent walk over tiles: virtual block vb = blockIdx.x + k * gridDim.x (gridDim.x is a multiple of 8 or
  // equals ntiles, so vb & 7 is this block's XCD for every k).  XCD-aware order: the virtual blocks of one XCD cover
  // a contiguous range of tiles; channel tiles fastest, then x, y, image.
  int tb, ty0, tx0, tch;   // image, first row, first column, first channel of the CURRENT tile
  auto decode = [&](int vb) __attribute__((always_inline)) {
    const int xcd = vb & 7, q = ntiles >> 3, r = ntiles & 7;
    const int L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (vb >> 3);
    const int tile_ch = L % tiles_ch;
    int rest = L / tiles_ch;
    const int tx = rest % tiles_x;
    rest /= tiles_x;
    const int ty = rest % tiles_y;
    tb = rest / tiles_y;
    tch = tile_ch * BCH;
    ty0 = ty * TH;
    tx0 = tx * TS;
  };

  // ---- per-tile loader state
  const char* xlane;       // this lane's patch source at tap 0, chunk 0 (may point outside the image: see pvalid)
  int iy0;                 // image row of that source
  unsigned hlim;           // rows iy with (unsigned)iy < hlim are loadable (inside the image and inside the patch)
  bool col_ok;             // this lane's patch column is inside the image
  const char* wtile;       // weight matrix of the tile's channel block (uniform)
  auto setup = [&]() __attribute__((always_inline)) {
    iy0 = ty0 - 1 + pdy;
    const int ix = tx0 - 1 + ppx;
    col_ok = ppx < TS + 2 && (unsigned)ix < (unsigned)W;
    const int hl = ty0 - 1 + PH;
    hlim = (unsigned)(hl < H ? hl : H);
    xlane = (const char*)(a.x + (long)tb * a.x_bstride + ((long)iy0 * W + ix) * a.ldx + pcc * 8);
    wtile = (const char*)(a.w + (long)tch * a.Kpad);
  };
  // tap t of chunk c: LDS rows 40 t + [0, 48) of buffer c & 1
  auto issue_patch_piece = [&](int c, int t) __attribute__((always_inline)) {
    const bool ok = col_ok && (unsigned)(iy0 + 2 * t) < hlim;
    const char* src = ok ? xlane + ((long)t * prow_bytes + c * 64) : (const char*)a.zero;
    glds16(src, smem + (c & 1) * PATCH_BYTES + t * (40 * ROWB) + wave * 1024);
  };
  // weights of (chunk c, tap t) into ring slot `slot`; koff2 = byte offset of that K slice inside a weight row
  auto issue_weights = [&](int koff2, int slot) __attribute__((always_inline)) {
    const char* w0 = wtile + koff2;
#pragma unroll
    for (int i = 0; i < W_IT; ++i) glds16(w0 + i * wblk + (unsigned)wlane, wbase + slot * WBUF + (i * 64 + wave * 16) * ROWB);
  };
  // tile prologue: bias (oldest DMA, so every counted wait covers it), patch of chunk 0, weights of row 0.
  // Exactly (NWB - 1) * W_IT of these are younger than stage 0.
  auto issue_prologue = [&](int par) __attribute__((always_inline)) {
    if (wave < BCH / 64) glds4(a.bias + tch + wave * 64 + lane, sbias + par * BCH + wave * 64);
    if (wave < 3) {
#pragma unroll
      for (int t = 0; t < NTP; ++t) issue_patch_piece(0, t);
    }
#pragma unroll
    for (int kw = 0; kw < NWB; ++kw) issue_weights(kw * a.Cin * 2, kw);
  };

  float4v acc[MT][NT];
  half8 af0[MT], af1[MT], bf[NT];
  // epilogue of a finished tile (bias from LDS, SiLU, residual, fp16 pack, 16-byte stores at a channel offset)
  auto epilogue = [&](int eb, int ey0, int ex0, int ech, int par) __attribute__((always_inline)) {
    // fast path: this wave's 8 rows x 16 columns x 64 channels are all inside the tensor
    if (ey0 + wpx * NT + NT <= H && ex0 + TS <= W && ech + wch * 64 + 64 <= a.Cout && !(a.dbg & (12 | 256))) {
      const float* sb = sbias + par * BCH + wch * 64 + g * 8;
      float4v bv[MT / 2][2];
#pragma unroll
      for (int sg = 0; sg < MT / 2; ++sg) {
        bv[sg][0] = *(const float4v*)(sb + sg * 32);
        bv[sg][1] = *(const float4v*)(sb + sg * 32 + 4);
      }
      const long pix0 = (long)(ey0 + wpx * NT) * W + ex0 + l15;
      const int cho = ech + wch * 64 + g * 8;
      half_t* yp = (half_t*)a.y + (long)eb * a.y_bstride + pix0 * a.ldy + cho;
      const long ystep = (long)W * a.ldy;
      if (a.res) {
        const half_t* rp = a.res + (long)eb * a.r_bstride + pix0 * a.ldr + cho;
        const long rstep = (long)W * a.ldr;
        if (a.act) conv_epilogue_fast<MT, NT, true, true>(acc, bv, yp, ystep, rp, rstep);
        else conv_epilogue_fast<MT, NT, false, true>(acc, bv, yp, ystep, rp, rstep);
      } else {
        if (a.act) conv_epilogue_fast<MT, NT, true, false>(acc, bv, yp, ystep, nullptr, 0);
        else conv_epilogue_fast<MT, NT, false, false>(acc, bv, yp, ystep, nullptr, 0);
      }
      return;
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int iy = ey0 + wpx * NT + nt, ix = ex0 + l15;
      if (iy >= H || ix >= W) continue;
      const long pix = (long)iy * W + ix;
#pragma unroll
      for (int sg = 0; sg < MT / 2; ++sg) {
        const int chl = wch * 64 + sg * 32 + g * 8;
        const int ch0 = ech + chl;
        if (ch0 >= a.Cout) continue;
        const float4v b0 = *(const float4v*)(sbias + par * BCH + chl), b1 = *(const float4v*)(sbias + par * BCH + chl + 4);
        float v[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          v[j] = acc[2 * sg][nt][j] + b0[j];
          v[4 + j] = acc[2 * sg + 1][nt][j] + b1[j];
        }
        if (a.act && !(a.dbg & 4)) {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = silu_f(v[j]);
        }
        if (a.res) {
          const half8 rv = *(const half8*)(a.res + (long)eb * a.r_bstride + pix * a.ldr + ch0);
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] += (float)rv[j];
        }
        if ((a.dbg & 8) && v[0] != 123.f) continue;  // dbg: no stores
        half8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = m355_to_half(v[j]);
        *(half8*)((half_t*)a.y + (long)eb * a.y_bstride + pix * a.ldy + ch0) = o;
      }
    }
  };

  const int nrows = nchunks * 3;   // K rows (chunk, kh); even because Cin is a multiple of 64
  int vb = blockIdx.x;
  decode(vb);
  setup();
  issue_prologue(0);
  int par = 0, ntile = 0;
  int pb_ = 0, py0_ = 0, px0_ = 0, pch_ = 0;   // previous tile (its accumulators are still in registers)
  bool have_prev = false;
  for (;;) {
    // The finished tile's epilogue runs while this tile's prologue DMA is in flight.  Its stores are the youngest
    // vector-memory operations, so the counted wait below lets all 16 of them (full tile) stay outstanding.
    bool prev_full = false;
    if (a.stamps && ntile == 1) sa = __builtin_amdgcn_s_memtime();
    if (a.stamps && ntile == 2) sa2 = __builtin_amdgcn_s_memtime();
    if (have_prev) {
      prev_full = (py0_ + wpx * NT + NT <= H) && (pch_ + wch * 64 + 64 <= a.Cout) && !(a.dbg & 8);
      epilogue(pb_, py0_, px0_, pch_, par ^ 1);
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = float4v{0.f, 0.f, 0.f, 0.f};
    if (a.stamps && ntile == 1) sb = __builtin_amdgcn_s_memtime();
    if (prev_full)
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((NWB - 1) * W_IT + 2 * NT) : "memory");
    else
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((NWB - 1) * W_IT) : "memory");  // patch 0 + stage 0 landed
    __builtin_amdgcn_s_barrier();

    // ---- row state: (chunk, kh) of the current row, its patch buffer and B base; the same for the next row
    int chunk = 0, kh = 0;
    int pbuf = 0, pbk = pb;
    int be = pbuf + (pbk << 6) + (g16 ^ ((pbk & 4) << 3));
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) af0[mt] = *(const half8*)(wbase + aoff + mt * 1024);
#pragma unroll
    for (int nt = 0; nt < HALF; ++nt) bf[nt] = *(const half8*)(smem + (be ^ ((nt & 1) << 5)) + nt * (PP * ROWB));

    if (a.stamps && ntile == 0) st1 = __builtin_amdgcn_s_memtime();
    if (a.stamps && ntile == 1) sc = __builtin_amdgcn_s_memtime();

#define M355_SB __builtin_amdgcn_sched_barrier(0);
#define M355_MF(AC, mt, nt) acc[(mt)][(nt)] = __builtin_amdgcn_mfma_f32_16x16x32_f16(AC[(mt)], bf[(nt)], acc[(mt)][(nt)], 0, 0, 0);
#define M355_RB(nt) bf[(nt)] = *(const half8*)(smem + (be ^ (((nt) & 1) << 5)) + (nt) * (PP * ROWB));
#define M355_RA(AN, i) AN[(i)] = *(const half8*)(wn + (i) * 1024);
    // One K step with compile-time kw.  The instruction order is pinned with sched_barrier(0) after every
    // (ds_read, MFMA) pair: left to itself the scheduler clusters the reads and idles the MFMA pipe.
    //   P1 : 4 HALF MFMA (image rows 0 .. HALF-1), the B reads of rows HALF .. NT-1 in the first shadows
    //   mid: stage s+1 landed (counted vmcnt: only the weights issued in the previous step may be in flight), barrier
    //   P2a: patch piece of tap 3 kh + kw for the next chunk, weights of (next row, kw) into slot kw
    //   P2b: 4 HALF MFMA (rows HALF .. NT-1) with the reads of step s+1 (A, then B rows 0 .. HALF-1) between them
#define M355_LEAN_STEP(AC, AN, KW)                                                                               \
  {                                                                                                              \
    _Pragma("unroll") for (int i = 0; i < 4 * HALF; ++i) {                                                       \
      if (i >= 1 && i <= HALF) { M355_RB(HALF + i - 1) }                                                         \
      M355_MF(AC, i & 3, i >> 2)                                                                                 \
      if (i <= HALF || i == 4 * HALF - 1) { M355_SB }                                                            \
    }                                                                                                            \
    if (lastrow)                                                                                                 \
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                                \
    else                                                                                                         \
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(W_IT) : "memory");                                     \
    __builtin_amdgcn_s_barrier();                                                                                \
    if (do_p && wave < 3 && 3 * kh + KW < NTP) issue_patch_piece(chunk + 1, 3 * kh + KW);                        \
    if (!lastrow) issue_weights(wnext + KW * cin2, KW);                                                          \
    {                                                                                                            \
      const int p = (KW < 2 ? pbk : pbk_n) + (KW + 1) % 3;                                                       \
      be = (KW < 2 ? pbuf : pbuf_n) + (p << 6) + (g16 ^ ((p & 4) << 3));                                         \
    }                                                                                                            \
    const char* wn = wbase + ((KW + 1) % 3) * WBUF + aoff;                                                       \
    M355_SB                                                                                                      \
    _Pragma("unroll") for (int i = 0; i < 4 * HALF; ++i) {                                                       \
      if (i >= 1 && i <= 4) { M355_RA(AN, i - 1) }                                                               \
      if (i >= 5 && i <= 4 + HALF) { M355_RB(i - 5) }                                                            \
      M355_MF(AC, i & 3, HALF + (i >> 2))                                                                        \
      if (i <= 4 + HALF || i == 4 * HALF - 1) { M355_SB }                                                        \
    }                                                                                                            \
  }
    // per-row scalars: the next row (chunk_n, kh_n), its B base, the weight offset of the next row, prefetch flags
#define M355_ROW_BEGIN                                                                                           \
    int kh_n = kh + 1, chunk_n = chunk;                                                                          \
    if (kh_n == 3) { kh_n = 0; ++chunk_n; }                                                                      \
    const int pbuf_n = (chunk_n & 1) * PATCH_BYTES;                                                              \
    const int pbk_n = pb + kh_n * PP;                                                                            \
    const int wnext = (3 * kh_n * a.Cin + chunk_n * 32) * 2;                                                     \
    const bool lastrow = row + 1 >= nrows;                                                                       \
    const bool do_p = chunk + 1 < nchunks && !(a.dbg & 1);
#define M355_ROW_END                                                                                             \
    kh = kh_n; chunk = chunk_n; pbuf = pbuf_n; pbk = pbk_n; ++row;

    const int cin2 = a.Cin * 2;
    for (int row = 0; row < nrows;) {
      {
        M355_ROW_BEGIN
        M355_LEAN_STEP(af0, af1, 0) M355_LEAN_STEP(af1, af0, 1) M355_LEAN_STEP(af0, af1, 2)
        M355_ROW_END
      }
      {
        M355_ROW_BEGIN
        M355_LEAN_STEP(af1, af0, 0) M355_LEAN_STEP(af0, af1, 1) M355_LEAN_STEP(af1, af0, 2)
        M355_ROW_END
      }
    }
#undef M355_LEAN_STEP
#undef M355_ROW_BEGIN
#undef M355_ROW_END
#undef M355_SB
#undef M355_MF
#undef M355_RB
#undef M355_RA

    if (a.stamps && ntile == 0) st2 = __builtin_amdgcn_s_memtime();
    if (a.stamps && ntile == 1) sd = __builtin_amdgcn_s_memtime();
    // After the last step's barrier no wave reads live LDS data any more (its second half only pre-reads the
    // never-used step after the end), so the next tile's DMA may start without another barrier.
    pb_ = tb; py0_ = ty0; px0_ = tx0; pch_ = tch;
    have_prev = true;
    par ^= 1;
    ++ntile;
    vb += gridDim.x;
    if (vb >= ntiles) break;
    decode(vb);
    setup();
    issue_prologue(par);
  }
  epilogue(pb_, py0_, px0_, pch_, par ^ 1);
  if (a.stamps && tid == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long st3 = __builtin_amdgcn_s_memtime();
    unsigned long long* o = a.stamps + (long)blockIdx.x * 8;  // [0..2]: first tile; [3]: block end; [6]: tiles done
    o[0] = st0; o[1] = st1; o[2] = st2; o[3] = st3; o[4] = rt0; o[5] = __builtin_amdgcn_s_memrealtime();
    o[6] = (unsigned long long)ntile;
    unsigned long long* o2 = a.stamps + (1 << 19) + (long)blockIdx.x * 4;  // second tile: epilogue start / end, loop start / end
    o2[0] = sa; o2[1] = sb; o2[2] = sc; o2[3] = sd;
    a.stamps[(1 << 19) + (1 << 18) + blockIdx.x] = sa2;
  }
}

template <int BCH, int TH>
int launch_lean(const ConvArgs& a, hipStream_t s) {
  using L = Lean<BCH, TH>;
  if (!conv_rows_covered(a, BCH)) return -1;
  const int tiles_x = (a.Wi + TS - 1) / TS, tiles_y = (a.Hi + TH - 1) / TH;
  const int tiles_ch = (a.Cout + BCH - 1) / BCH;
  const int B = a.M / (a.Ho * a.Wo);
  const int ntiles = B * tiles_y * tiles_x * tiles_ch;
  static int slots = 0;   // resident blocks
  if (!slots) {
    hipError_t e = hipFuncSetAttribute((const void*)conv3x3_lean_kernel<BCH, TH>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       L::LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
      return -2;
    slots = (L::BLOCKS * cus) & ~7;   // the XCD-aware tile order needs gridDim.x % 8 == 0 whenever a block walks > 1 tile
    if (slots < 8) slots = 8;
  }
  const int grid = ntiles <= slots ? ntiles : slots;
  hipLaunchKernelGGL((conv3x3_lean_kernel<BCH, TH>), dim3(grid), dim3(256), L::LDS_BYTES, s, a, tiles_x, tiles_y, a.Cin / 32,
                     ntiles);
  return (int)hipGetLastError();
}

}  // namespace

// which: 0 = by shape (64-channel block when Cout <= 64, else 128 ch x 8 rows), 1 = <128, 8>, 2 = <64, 16>, 3 = <128, 16>
int launch_conv3x3_lean(const ConvArgs& a, int which, hipStream_t s) {
  if (!conv3x3_halo_ok(a)) return -1;
  if (which == 0) which = a.Cout <= 64 ? 2 : 1;
  if (which == 1) return launch_lean<128, 8>(a, s);
  if (which == 2) return launch_lean<64, 16>(a, s);
  if (which == 3) return launch_lean<128, 16>(a, s);
  return -1;
}

}  // namespace m355
