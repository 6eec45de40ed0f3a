// 3x3 / stride-1 / pad-1 NHWC fp16 convolution for NARROW maps (W <= 26: the 20x20 level at 640x640 input), where a
// 16-pixel-wide tile wastes 38-60 % of its columns and the layers fell back to the im2col kernel (which re-streams the
// input nine times through the 30 B/clk/CU LDS intake: 16 % MFMA share on 256 -> 256 @ 20x20).
// Same pipeline as the wide kernel's small-tile form (32-deep K steps in rows of three taps, weight ring slot = kw, affine patch pieces
// through a zero page, persistent tile walk, epilogue under the next prologue), different geometry:
//   * a tile is a SLAB of R full-width image rows (R x W <= 256 pixels, R <= 10) x 64 channels; the MFMA pixel groups
//     are 16 CONSECUTIVE pixels of the slab in row-major order, so a group may straddle two image rows and no column
//     is wasted (20x20: two slabs of 10 rows = 200 pixels = 12.5 groups per image and channel block);
//   * four waves x (64 channels x 4 pixel groups); each group keeps the LDS row of its lane's pixel at tap (0, 0) in
//     a register and adds the tap offset once per step (the 16x16 tiles get that for free from the row pitch);
//   * patch rows are W + 2 <= 28 pixels at a pitch of 28 LDS rows: conflict-free for these straddling 16-lane reads
//     with the usual 64-byte-row swizzle (brute-forced), and 2 patch rows = 56 LDS rows = one piece per wave and tap.
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

constexpr int PP = 28;                // patch row pitch in LDS rows (W + 2 <= 28 used)
constexpr int ROWB = 64;              // LDS row = 32 halves
constexpr int NWB = 6;                // weight ring, two K rows deep: the slot of step (row, kw) is 3 (row & 1) + kw
constexpr int MT = 4;

constexpr int BCH = 64;
constexpr int NT = 4;                 // pixel groups per wave
constexpr int NTP_MAX = 6;            // taps that may carry a patch piece: 12 patch rows = slabs of up to 10 image rows
constexpr int PROWS = 56 * (NTP_MAX - 1) + 64;     // 344
constexpr int PATCH_BYTES = PROWS * ROWB;          // 22016
constexpr int WBUF = BCH * ROWB;                   // 4096
constexpr int W_IT = 1;
constexpr int LDS_BYTES = 2 * PATCH_BYTES + NWB * WBUF + 2 * BCH * 4;   // 69120: two blocks per CU
constexpr int HALF = NT / 2;

__global__ __launch_bounds__(256, 2) void conv3x3_slab_kernel(const ConvArgs a, int R, int slabs, int nchunks, int ntiles) {
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
  const int wch = 0, wpx = wave;
  const int l15 = lane & 15, g = lane >> 4;
  const int aoff = (wch * 64 + l15) * ROWB + ((g ^ (((l15 >> 2) & 1) << 1)) << 4);   // + mt * 1024 (immediate)
  // pixel groups of this wave: group index 4 wave + nt, pixel p = 16 group + l15 of the slab in row-major order
  const int npx = R * W;                              // pixels of a full slab
  const int ntp = (R + 3) >> 1;                       // taps that carry a patch piece: ceil((R + 2) / 2)
  int pbase[NT];                                      // LDS row of this lane's pixel at tap (0, 0)
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    int p = 16 * (wpx * NT + nt) + l15;
    p = p < npx ? p : npx - 1;                        // lanes / groups past the slab recompute its last pixel (never stored)
    const int y = p / W;
    pbase[nt] = y * PP + (p - y * W);
  }
  const int g16 = g << 4;

  // ---- patch streaming lane constants (all four waves): LDS row 56 t + r0 at tap t
  const int r0 = wave * 16 + lrow;
  const int pdy = r0 / PP, ppx = r0 - pdy * PP;
  const int pcc = lslot ^ (((r0 >> 2) & 1) << 1);                 // (56 t + r0) >> 2 has the parity of r0 >> 2
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
  int tb, ty0, tch;   // image, first row, first channel of the CURRENT tile
  auto decode = [&](int vb) __attribute__((always_inline)) {
    const int xcd = vb & 7, q = ntiles >> 3, r = ntiles & 7;
    const int L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (vb >> 3);
    const int tile_ch = L % tiles_ch;
    const int rest = L / tiles_ch;
    const int slab = rest % slabs;
    tb = rest / slabs;
    tch = tile_ch * BCH;
    ty0 = slab * R;
  };

  // ---- per-tile loader state
  const char* xlane;       // this lane's patch source at tap 0, chunk 0 (may point outside the image: see pvalid)
  int iy0;                 // image row of that source
  unsigned hlim;           // rows iy with (unsigned)iy < hlim are loadable (inside the image and inside the patch)
  bool col_ok;             // this lane's patch column is inside the image
  const char* wtile;       // weight matrix of the tile's channel block (uniform)
  auto setup = [&]() __attribute__((always_inline)) {
    iy0 = ty0 - 1 + pdy;
    const int ix = ppx - 1;
    col_ok = ppx < W + 2 && (unsigned)ix < (unsigned)W;
    const int hl = ty0 + R + 1;
    hlim = (unsigned)(hl < H ? hl : H);
    xlane = (const char*)(a.x + (long)tb * a.x_bstride + ((long)iy0 * W + ix) * a.ldx + pcc * 8);
    wtile = (const char*)(a.w + (long)tch * a.Kpad);
  };
  // tap t of chunk c: LDS rows 56 t + [0, 64) of buffer c & 1
  auto issue_patch_piece = [&](int c, int t) __attribute__((always_inline)) {
    const bool ok = col_ok && (unsigned)(iy0 + 2 * t) < hlim;
    const char* src = ok ? xlane + ((long)t * prow_bytes + c * 64) : (const char*)a.zero;
    glds16(src, smem + (c & 1) * PATCH_BYTES + t * (56 * ROWB) + wave * 1024);
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
    if (wave == 0) glds4(a.bias + tch + lane, sbias + par * BCH);
    // always NTP_MAX pieces, so that the counted waits below hold for every R (pieces past the slab read the zero page)
#pragma unroll
    for (int t = 0; t < NTP_MAX; ++t) issue_patch_piece(0, t);
#pragma unroll
    for (int j = 0; j < NWB; ++j) issue_weights(j * a.Cin * 2, j);   // rows (chunk 0, kh 0) and (chunk 0, kh 1): taps 0 .. 5
  };

  float4v acc[MT][NT];
  half8 af0[MT], af1[MT], bf[NT];
  // epilogue of a finished tile (bias from LDS, SiLU, residual, fp16 pack, 16-byte stores at a channel offset).
  // The slab's pixels are consecutive in the NHWC image, so group nt of this wave starts 16 pixels after group nt - 1.
  auto wave_full = [&](int ey0, int ech) __attribute__((always_inline)) {
    return 16 * (wpx * NT + NT) <= npx && ey0 + R <= H && ech + 64 <= a.Cout && !(a.dbg & (12 | 256));
  };
  auto epilogue = [&](int eb, int ey0, int ech, int par) __attribute__((always_inline)) {
    const long pix_w = (long)ey0 * W + 16 * (wpx * NT) + l15;     // image-linear pixel of group 0 of this wave
    if (wave_full(ey0, ech)) {
      const float* sb = sbias + par * BCH + g * 8;
      float4v bv[MT / 2][2];
#pragma unroll
      for (int sg = 0; sg < MT / 2; ++sg) {
        bv[sg][0] = *(const float4v*)(sb + sg * 32);
        bv[sg][1] = *(const float4v*)(sb + sg * 32 + 4);
      }
      const int cho = ech + g * 8;
      half_t* yp = (half_t*)a.y + (long)eb * a.y_bstride + pix_w * a.ldy + cho;
      const long ystep = 16L * a.ldy;
      if (a.res) {
        const half_t* rp = a.res + (long)eb * a.r_bstride + pix_w * a.ldr + cho;
        const long rstep = 16L * a.ldr;
        if (a.act) conv_epilogue_fast<MT, NT, true, true>(acc, bv, yp, ystep, rp, rstep);
        else conv_epilogue_fast<MT, NT, false, true>(acc, bv, yp, ystep, rp, rstep);
      } else {
        if (a.act) conv_epilogue_fast<MT, NT, true, false>(acc, bv, yp, ystep, nullptr, 0);
        else conv_epilogue_fast<MT, NT, false, false>(acc, bv, yp, ystep, nullptr, 0);
      }
      return;
    }
    const int lim = (H - ey0 < R ? H - ey0 : R) * W;              // valid pixels of this slab
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int p = 16 * (wpx * NT + nt) + l15;
      if (p >= lim) continue;
      const long pix = (long)ey0 * W + p;
#pragma unroll
      for (int sg = 0; sg < MT / 2; ++sg) {
        const int chl = sg * 32 + g * 8;
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
  int pb_ = 0, py0_ = 0, pch_ = 0;   // previous tile (its accumulators are still in registers)
  bool have_prev = false;
  for (;;) {
    // The finished tile's epilogue runs while this tile's prologue DMA is in flight.  Its stores are the youngest
    // vector-memory operations, so the counted wait below lets all 16 of them (full tile) stay outstanding.
    bool prev_full = false;
    if (a.stamps && ntile == 1) sa = __builtin_amdgcn_s_memtime();
    if (a.stamps && ntile == 2) sa2 = __builtin_amdgcn_s_memtime();
    if (have_prev) {
      prev_full = wave_full(py0_, pch_);
      epilogue(pb_, py0_, pch_, par ^ 1);
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
    int pbuf = 0;
    int ba[NT];          // byte address of this lane's B fragment of each group for the step being read
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) ba[nt] = (pbase[nt] << 6) + (g16 ^ ((pbase[nt] & 4) << 3));
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) af0[mt] = *(const half8*)(wbase + aoff + mt * 1024);
#pragma unroll
    for (int nt = 0; nt < HALF; ++nt) bf[nt] = *(const half8*)(smem + ba[nt]);

    if (a.stamps && ntile == 0) st1 = __builtin_amdgcn_s_memtime();
    if (a.stamps && ntile == 1) sc = __builtin_amdgcn_s_memtime();

#define M355_SB __builtin_amdgcn_sched_barrier(0);
#define M355_MF(AC, mt, nt) acc[(mt)][(nt)] = __builtin_amdgcn_mfma_f32_16x16x32_f16(AC[(mt)], bf[(nt)], acc[(mt)][(nt)], 0, 0, 0);
#define M355_RB(nt) bf[(nt)] = *(const half8*)(smem + ba[(nt)]);
#define M355_RA(AN, i) AN[(i)] = *(const half8*)(wn + (i) * 1024);
    // One K step with compile-time kw.  The instruction order is pinned with sched_barrier(0) after every
    // (ds_read, MFMA) pair: left to itself the scheduler clusters the reads and idles the MFMA pipe.
    //   P1 : 4 HALF MFMA (image rows 0 .. HALF-1), the B reads of rows HALF .. NT-1 in the first shadows
    //   mid: stage s+1 landed (counted vmcnt: only the weights issued in the previous step may be in flight), barrier
    //   P2a: patch piece of tap 3 kh + kw for the next chunk, weights of (next row, kw) into slot kw
    //   P2b: 4 HALF MFMA (rows HALF .. NT-1) with the reads of step s+1 (A, then B rows 0 .. HALF-1) between them
#define M355_SLAB_STEP(AC, AN, KW, SC)                                                                              \
  {                                                                                                              \
    _Pragma("unroll") for (int i = 0; i < 4 * HALF; ++i) {                                                       \
      if (i >= 1 && i <= HALF) { M355_RB(HALF + i - 1) }                                                         \
      M355_MF(AC, i & 3, i >> 2)                                                                                 \
      if (i <= HALF || i == 4 * HALF - 1) { M355_SB }                                                            \
    }                                                                                                            \
    if (tail)                                                                                                    \
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                                \
    else                                                                                                         \
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(4 * W_IT) : "memory");                                 \
    __builtin_amdgcn_s_barrier();                                                                                \
    if (do_p && 3 * kh + KW < ntp) issue_patch_piece(chunk + 1, 3 * kh + KW);                                    \
    if (!tail) issue_weights(wnext + KW * cin2, SC + KW);                                                        \
    {                                                                                                            \
      const int tapo = (KW < 2 ? kh : kh_n) * PP + (KW + 1) % 3;                                                 \
      const int pbn = (KW < 2 ? pbuf : pbuf_n);                                                                  \
      _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) {                                                        \
        const int p = pbase[nt] + tapo;                                                                          \
        ba[nt] = pbn + (p << 6) + (g16 ^ ((p & 4) << 3));                                                        \
      }                                                                                                          \
    }                                                                                                            \
    const char* wn = wbase + ((KW < 2 ? SC : 3 - SC) + (KW + 1) % 3) * WBUF + aoff;                              \
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
    int kh_2 = kh + 2, chunk_2 = chunk;                                                                          \
    if (kh_2 >= 3) { kh_2 -= 3; ++chunk_2; }                                                                     \
    const int wnext = (3 * kh_2 * a.Cin + chunk_2 * 32) * 2;     /* the row after the next */                    \
    const bool tail = row + 2 >= nrows;                          /* its weights do not exist: nothing left in flight */ \
    const bool do_p = chunk + 1 < nchunks && !(a.dbg & 1);
#define M355_ROW_END                                                                                             \
    kh = kh_n; chunk = chunk_n; pbuf = pbuf_n; ++row;

    const int cin2 = a.Cin * 2;
    for (int row = 0; row < nrows;) {
      {
        M355_ROW_BEGIN
        M355_SLAB_STEP(af0, af1, 0, 0) M355_SLAB_STEP(af1, af0, 1, 0) M355_SLAB_STEP(af0, af1, 2, 0)
        M355_ROW_END
      }
      {
        M355_ROW_BEGIN
        M355_SLAB_STEP(af1, af0, 0, 3) M355_SLAB_STEP(af0, af1, 1, 3) M355_SLAB_STEP(af1, af0, 2, 3)
        M355_ROW_END
      }
    }
#undef M355_SLAB_STEP
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
    pb_ = tb; py0_ = ty0; pch_ = tch;
    have_prev = true;
    par ^= 1;
    ++ntile;
    vb += gridDim.x;
    if (vb >= ntiles) break;
    decode(vb);
    setup();
    issue_prologue(par);
  }
  epilogue(pb_, py0_, pch_, par ^ 1);
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

// Slab height for a map: R rows with R x W <= 256 pixels (16 groups = 4 waves x 4) and R + 2 <= 12 patch rows.
static int slab_rows(int H, int W) {
  int R = 256 / W;
  if (R > 10) R = 10;
  if (R > H) R = H;
  return R;
}

// Eligibility: 3x3 stride 1 pad 1, fp16 NHWC, Cin a multiple of 64, Cout >= 64, map at most 26 pixels wide.
bool conv3x3_slab_ok(const ConvArgs& a) {
  if (a.ksize != 3 || a.stride != 1 || a.pad != 1 || a.out_f32 || a.convt_co > 0) return false;
  if (a.Cin % 64 || a.Cout < 64 || a.Cout % 8 || a.ldx % 8 || a.ldy % 8) return false;
  if (a.Ho != a.Hi || a.Wo != a.Wi) return false;
  return a.Wi >= 1 && a.Wi <= PP - 2 && a.Hi >= 1;
}

int launch_conv3x3_slab(const ConvArgs& a, hipStream_t s) {
  if (!conv3x3_slab_ok(a) || !conv_rows_covered(a, BCH)) return -1;
  const int R = slab_rows(a.Hi, a.Wi);
  const int slabs = (a.Hi + R - 1) / R;
  const int tiles_ch = (a.Cout + BCH - 1) / BCH;
  const int B = a.M / (a.Ho * a.Wo);
  const int ntiles = B * slabs * tiles_ch;
  static int slots = 0;   // resident blocks: two per CU
  if (!slots) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
      return -2;
    slots = (2 * cus) & ~7;   // the XCD-aware tile order needs gridDim.x % 8 == 0 whenever a block walks > 1 tile
    if (slots < 8) slots = 8;
  }
  const int grid = ntiles <= slots ? ntiles : slots;
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute((const void*)conv3x3_slab_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    attr = true;
  }
  hipLaunchKernelGGL(conv3x3_slab_kernel, dim3(grid), dim3(256), LDS_BYTES, s, a, R, slabs, a.Cin / 32, ntiles);
  return (int)hipGetLastError();
}

}  // namespace m355
