// 3x3 / stride-1 / pad-1 NHWC fp16 convolution, WIDE halo tile: 128 channels x (16 x 16) pixels per workgroup,
// four waves, each owning 64 channels x 8 image rows (MT=4 x NT=8 MFMA tiles, 128 fp32 accumulators per lane).
//
// Why a third conv kernel.  Stamps on conv3x3_halo (128 ch x 8x16 px, tools/stamps_halo.py, profiles/r01_*):
// of a block's 45 k cycles 12 % are prologue and 23 % epilogue; in the loop the 2 x 9216 MFMA cycles per SIMD are
// 64 % of the wall, and switching the in-loop LDS-DMA off brings that to 84 % -- every 1 KiB DMA piece costs the
// issuing wave 100+ cycles next to MFMAs, and each step streams 16 KiB of weights for 32 MFMAs per wave.
// This kernel doubles the pixels per weight byte and halves the K depth per step (32 instead of 64 channels):
//   per step and wave: 32 MFMAs again, but 2 weight pieces + <=1 patch piece (was 4 + 1), 12 ds_read_b128 (was 16);
//   per block: 36 instead of 18 steps behind one prologue / epilogue (Cin = 128).
// LDS (77.5 KiB, two blocks per CU): two patch buffers of 18 rows x 20 px x 64 B (px 18, 19 are padding: a row
// pitch of 20 makes the swizzle of image row nt+1 the complement of row nt, so the eight B-fragment reads of a
// tap share two address registers and use the instruction's immediate offset), a 4-deep ring of 8 KiB weight
// stages, 512 B of bias.  Rows are 64 B = four 16-byte chunks; chunk c of row r lives in slot c ^ (2 * ((r >> 2) & 1)):
// conflict-free for ds_read_b128 at every row offset (brute-forced over the four 16-lane service groups).
//
// Pipeline (one barrier per step, mid-step; A fragments double-buffered, B fragments single-buffered in halves):
//   P1(s): 8 MFMA (rows 0-1) | ds_read B rows 4-7 of step s | 8 MFMA (rows 2-3)
//   wait : stage s+1 landed (counted vmcnt), reads landed, s_barrier
//   P2(s): 8 MFMA (rows 4-5) | LDS-DMA weights of step s+4 (+ one patch piece of the next chunk),
//          ds_read A(s+1), B rows 0-3 of step s+1 | 8 MFMA (rows 6-7)
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

constexpr int TS = 16, TH = 16;       // output tile
constexpr int PP = 20;                // patch row pitch in pixels (18 used)
constexpr int PH = TH + 2;
constexpr int PROWS = 368;            // 18 x 20 patch pixels + 8 rows the last DMA piece spills into (never read)
constexpr int ROWB = 64;              // LDS row = 32 halves
constexpr int PATCH_BYTES = PROWS * ROWB;          // 23552
constexpr int BCH = 128;
constexpr int WBUF = BCH * ROWB;                   // 8192
constexpr int NWB = 3;                             // weight ring: the slot of a step is its kw
constexpr int W_IT = BCH / (4 * 16);               // 2 weight pieces per wave per step
constexpr int MT = 4, NT = 8;
constexpr int LDS_BYTES = 2 * PATCH_BYTES + NWB * WBUF + 2 * BCH * 4;

// Instruction economy is the design rule of this kernel.  A wave issues one instruction per four cycles, and 32 MFMAs
// of 16 cycles leave 96 issue slots per step for everything else; the first version spent ~170 (a six-way EXEC-masked
// chain selecting the patch piece, 64-bit address adds, tap / chunk wrap arithmetic) and a wave alone needed 1,230
// cycles per 512-cycle step.  Here:
//   * K runs in ROWS of three steps (one kh, kw = 0, 1, 2 as compile-time constants): the weight ring slot is kw,
//     the weights of row r+1 / step kw are fetched at row r / step kw, all per-row scalars are computed once per row;
//   * the patch of the next chunk streams in 9 affine pieces: at tap t waves 0-2 load LDS rows 40 t .. 40 t + 47
//     (= patch rows 2t, 2t+1 and the first 8 pixels of 2t+2, rewritten identically by tap t+1), source offset
//     = lane base + t * 2 W ldx; out-of-image pixels read a zero page instead of being EXEC-masked.
__global__ __launch_bounds__(256, 2) void conv3x3_wide_kernel(const ConvArgs a, int tiles_x, int tiles_y, int nchunks,
                                                              int ntiles, int stagger) {
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
  const int wch = wave >> 1, wpx = wave & 1;
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
    col_ok = ppx < PH && (unsigned)ix < (unsigned)W;
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
    if (wave < 2) glds4(a.bias + tch + wave * 64 + lane, sbias + par * BCH + wave * 64);
    if (wave < 3) {
      for (int t = 0; t < 9; ++t) issue_patch_piece(0, t);
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
  // Two blocks share a CU and would run in lockstep (same tile size): their epilogues (512 quarter-rate exp / rcp per
  // lane, no MFMA) would coincide and their main loops would fight for the MFMA pipe.  The block whose LDS allocation
  // does not start at 0 is the CU's second one: it starts late, so one block's epilogue runs under the other's loop.
  if (stagger > 0 && (__builtin_amdgcn_s_getreg((6 /*HW_REG_LDS_ALLOC*/) | (0 << 6) | (11 << 11)) != 0)) {
    for (int i = 0; i < stagger; ++i) __builtin_amdgcn_s_sleep(127);
  }
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
    for (int nt = 0; nt < NT / 2; ++nt) bf[nt] = *(const half8*)(smem + (be ^ ((nt & 1) << 5)) + nt * (PP * ROWB));

    if (a.stamps && ntile == 0) st1 = __builtin_amdgcn_s_memtime();
    if (a.stamps && ntile == 1) sc = __builtin_amdgcn_s_memtime();

#define M355_SB __builtin_amdgcn_sched_barrier(0);
#define M355_MF(AC, mt, nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(AC[mt], bf[nt], acc[mt][nt], 0, 0, 0);
#define M355_RB(nt) bf[nt] = *(const half8*)(smem + (be ^ ((nt & 1) << 5)) + nt * (PP * ROWB));
#define M355_RA(AN, i) AN[i] = *(const half8*)(wn + i * 1024);
#define M355_ROW(AC, nt) M355_MF(AC, 0, nt) M355_MF(AC, 1, nt) M355_MF(AC, 2, nt) M355_MF(AC, 3, nt)
    // One K step with compile-time kw.  The instruction order is pinned with sched_barrier(0) after every
    // (ds_read, MFMA) pair: left to itself the scheduler clusters the reads and idles the MFMA pipe.
    //   P1 : 16 MFMA (image rows 0-3), the four B reads of rows 4-7 in the first shadows
    //   mid: stage s+1 landed (counted vmcnt: only the weights issued in the previous step may be in flight), barrier
    //   P2a: patch piece of tap 3 kh + kw for the next chunk, weights of (next row, kw) into slot kw
    //   P2b: 16 MFMA (rows 4-7) with the eight reads of step s+1 (A, then B rows 0-3) between them
#define M355_WIDE_STEP(AC, AN, KW)                                                                               \
  {                                                                                                              \
    M355_MF(AC, 0, 0) M355_SB M355_RB(4) M355_MF(AC, 1, 0) M355_SB M355_RB(5) M355_MF(AC, 2, 0) M355_SB           \
    M355_RB(6) M355_MF(AC, 3, 0) M355_SB M355_RB(7) M355_MF(AC, 0, 1) M355_SB                                    \
    M355_MF(AC, 1, 1) M355_MF(AC, 2, 1) M355_MF(AC, 3, 1) M355_ROW(AC, 2) M355_ROW(AC, 3) M355_SB                \
    if (lastrow)                                                                                                 \
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                                \
    else                                                                                                         \
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(W_IT) : "memory");                                     \
    __builtin_amdgcn_s_barrier();                                                                                \
    if (do_p && wave < 3) issue_patch_piece(chunk + 1, 3 * kh + KW);                                             \
    if (!lastrow) issue_weights(wnext + KW * cin2, KW);                                                          \
    {                                                                                                            \
      const int p = (KW < 2 ? pbk : pbk_n) + (KW + 1) % 3;                                                       \
      be = (KW < 2 ? pbuf : pbuf_n) + (p << 6) + (g16 ^ ((p & 4) << 3));                                         \
    }                                                                                                            \
    const char* wn = wbase + ((KW + 1) % 3) * WBUF + aoff;                                                       \
    M355_SB                                                                                                      \
    M355_MF(AC, 0, 4) M355_SB                                                                                    \
    M355_RA(AN, 0) M355_MF(AC, 1, 4) M355_SB M355_RA(AN, 1) M355_MF(AC, 2, 4) M355_SB                            \
    M355_RA(AN, 2) M355_MF(AC, 3, 4) M355_SB M355_RA(AN, 3) M355_MF(AC, 0, 5) M355_SB                            \
    M355_RB(0) M355_MF(AC, 1, 5) M355_SB M355_RB(1) M355_MF(AC, 2, 5) M355_SB                                    \
    M355_RB(2) M355_MF(AC, 3, 5) M355_SB M355_RB(3) M355_MF(AC, 0, 6) M355_SB                                    \
    M355_MF(AC, 1, 6) M355_MF(AC, 2, 6) M355_MF(AC, 3, 6) M355_ROW(AC, 7) M355_SB                                \
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
        M355_WIDE_STEP(af0, af1, 0) M355_WIDE_STEP(af1, af0, 1) M355_WIDE_STEP(af0, af1, 2)
        M355_ROW_END
      }
      {
        M355_ROW_BEGIN
        M355_WIDE_STEP(af1, af0, 0) M355_WIDE_STEP(af0, af1, 1) M355_WIDE_STEP(af1, af0, 2)
        M355_ROW_END
      }
    }
#undef M355_WIDE_STEP
#undef M355_ROW_BEGIN
#undef M355_ROW_END
#undef M355_SB
#undef M355_MF
#undef M355_RB
#undef M355_RA
#undef M355_ROW

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

}  // namespace

// Eligibility: as conv3x3_halo_ok, plus Cout >= 128 and at most 30 % of the computed pixels wasted by 16 x 16 tiles.
bool conv3x3_wide_ok(const ConvArgs& a) {
  if (!conv3x3_halo_ok(a) || a.Cout < 128) return false;
  const long covered = (long)((a.Hi + TH - 1) / TH) * TH * ((a.Wi + TS - 1) / TS) * TS;
  return covered * 10 <= (long)a.Hi * a.Wi * 13;
}

int launch_conv3x3_wide(const ConvArgs& a, hipStream_t s) {
  if (!conv3x3_wide_ok(a) || !conv_rows_covered(a, BCH)) return -1;
  const int tiles_x = (a.Wi + TS - 1) / TS, tiles_y = (a.Hi + TH - 1) / TH;
  const int tiles_ch = (a.Cout + BCH - 1) / BCH;
  const int B = a.M / (a.Ho * a.Wo);
  const int ntiles = B * tiles_y * tiles_x * tiles_ch;
  static int slots = 0;   // resident blocks: two per CU
  if (!slots) {
    hipError_t e = hipFuncSetAttribute((const void*)conv3x3_wide_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
      return -2;
    const char* ev = getenv("M355_WIDE_SLOTS");
    slots = ev ? atoi(ev) : 2 * cus;
    if (slots < 8) slots = 8;
    slots &= ~7;   // the XCD-aware tile order needs gridDim.x % 8 == 0 whenever a block walks more than one tile
  }
  const int grid = ntiles <= slots ? ntiles : slots;
  static int stagger = -1;
  if (stagger < 0) {
    const char* ev = getenv("M355_WIDE_STAGGER");
    stagger = ev ? atoi(ev) : 0;
  }
  hipLaunchKernelGGL(conv3x3_wide_kernel, dim3(grid), dim3(256), LDS_BYTES, s, a, tiles_x, tiles_y, a.Cin / 32, ntiles,
                     ntiles > grid ? stagger : 0);
  return (int)hipGetLastError();
}

}  // namespace m355
