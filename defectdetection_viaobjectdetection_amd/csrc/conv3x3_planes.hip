// 3x3 / stride-1 / pad-1 NHWC fp16 convolutions on v_mfma_f32_32x32x16_f16 over FULL-WIDTH ROW SLABS, with an optional second
// 3x3 convolution fed from LDS: a whole C2f Bottleneck (cv1 -> cv2 (+ shortcut)) in ONE launch (gfx950).
//
// Replaces: the Conv+BN+SiLU 3x3 pairs  m.cv1 -> m.cv2  of upstream's C2f Bottleneck (e = 1.0) that
// BscanBased/yolo8_seg_predict.py:8 reaches through torch.nn.functional.conv2d, two launches and one HBM round trip of the
// hidden tensor in rounds 1-3 (conv3x3_m32 / conv3x3_halo); in single-conv mode the 3x3 layers of the 20 x 20 level
// (conv3x3_small.hip's slab kernel).
//
// Why.  Round 3's stamps: a 128 -> 128 layer on a 40 x 40 map at batch 32 keeps the matrix pipe 42 % busy.  conv3x3_m32 runs
// two blocks per CU that each stream the whole 295 KB weight matrix through LDS for 128 pixels: 32 B/clk/CU of weight
// intake, which IS the L2 -> LDS intake limit (~30 B/clk/CU), plus a prologue, an epilogue and a half-empty tile round per
// launch.  Here ONE block per CU owns every output channel of a slab of R full-width image rows:
//   * a weight stage (one tap x 32 input channels x all output channels of the block) feeds 7-9 pixel blocks of 32 pixels per
//     wave instead of 2: 14-18 B/clk of weight intake;
//   * pixels are LINEAR in the slab at a pitch of W + 1: the zero column right of row y is the zero column left of row
//     y + 1, so a 40-pixel map wastes one column in 41 (the 8 x 16 tiles of m32 wasted 17 %), every MFMA pixel block is 32
//     consecutive storage indices and every tap is a constant index shift: fragment address = per-lane base(tap) + immediate;
//   * pair mode: the first convolution is evaluated on R + 2 rows (the halo rows of the second one are recomputed, 7 / 5 at
//     R = 5), goes through bias + SiLU + fp16 -- the rounding point of the two-launch form -- into LDS planes and never
//     reaches HBM; the second convolution reads it there; one launch, one prologue, one epilogue, no tail round between.
//
// Layout.  A PLANE is 32 channels of the slab: one 64-byte LDS row per storage index (pixel), 16-byte chunk c of row r at
// chunk position c ^ ((r >> 2) & 3).  A ds_read_b128 service group (16 lanes) covers 16 consecutive indices mod 16, so its
// rows hit all four (r & 3) bank quarters with four distinct chunk positions each: conflict-free for every tap shift, for the
// LDS-DMA's lane-linear writes (swizzle applied to the SOURCE address) and for the weight stages (same row format).
// GEMM orientation as in conv3x3_m32.hip: D[channel][pixel], weights = A operand, rows permuted on the DMA source side so
// that a lane-half owns 16 consecutive channels of its pixel (two 16-byte stores / LDS writes per pixel block).
//
// Block = 4 waves, one per SIMD (launch_bounds(256, 1): up to 512 VGPRs): wave = (channel block wc of WC, pixel group wp of
// WP = 4 / WC), NPB pixel blocks each (accumulators: 16 x NPB VGPRs).  K loop: phase = one input plane (32 channels), step =
// one tap = 2 MFMA slices of K = 16.  Weight ring of four stages: the stage of step g + 3 is issued in step g, a step waits
// (counted vmcnt) for the stage of step g + 1 only.  Input planes: ring of two; the pieces of plane p + 1 are issued during
// the first six steps of phase p.  ONE barrier per step.
#include <stdlib.h>

#include <algorithm>

#include "common.h"

namespace m355 {
namespace {

typedef float float16v __attribute__((ext_vector_type(16)));

constexpr int ROWB = 64;      // bytes per LDS row: 32 channels of one pixel / 32 K values of one weight row
constexpr int LDS_MAX = 160 * 1024;

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, int voff, int soff, char* lds) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds, 16, voff, soff, 0, 0);
}

struct PlanesGeom {
  int R, nslab, PW;         // output rows per slab, slabs per image, pitch W + 1
  int irows, xrows;         // rows of an intermediate plane (32 * NB1) / of an input ring slot (16 * npieces)
  int npieces;              // 1 KiB pieces of an input plane
  int NP, NPo;              // input planes (Cin / 32); planes of the hidden tensor (pair) = WC
  int tiles_ch;             // single mode: output channel tiles
  int ntiles;
  int off_x, off_s;         // LDS byte offsets: input ring, one spare KiB behind it (reads of never-stored pixel blocks may run past a slot)
};

template <int WC_, int NPB1_, int NPB2_, bool PAIR_, int PPS_>
struct PCfg {
  static constexpr int WC = WC_, WP = 4 / WC_, NPB1 = NPB1_, NPB2 = NPB2_, PPS = PPS_;
  static constexpr bool PAIR = PAIR_;
  static constexpr int NB1 = WP * NPB1, NB2 = WP * NPB2;
  static constexpr int NPB = NPB1 > NPB2 ? NPB1 : NPB2;
  static constexpr int PIT = 6 * PPS;             // input pieces per wave and plane (issued in steps 0 .. 5)
  static_assert(WC == 2 || WC == 4, "channel blocks per block");
};

template <class C>
struct PState {
  float16v acc[C::NPB];
  half8 wa[3][2];             // weight fragments (MFMA A operands) of three consecutive steps, straight from global memory
  half8 fb[2][C::NPB];
  int tb[9];                  // B fragment: LDS byte offset inside a plane for (tap, slice 0), pixel block 0 of this wave
  int pvoff[C::PIT];          // per-lane source offsets of the input pieces of the slab the loader is on
  __amdgpu_buffer_rsrc_t rs_x;
  char* smem;
  int wave, lane16;
  int pcur, pnext;            // LDS byte offset of the current / the next phase's plane
  // weight cursor: the fragments step g loads = those of step g + 2.  A convolution's fragments of one channel block are
  // contiguous in K-loop order (plane, tap, slice): the cursor is a byte offset that advances by 2 KiB per step.
  // weight stream: the fragments step g loads are those of step g + 2.  A convolution's fragments of one channel block are
  // contiguous in K-loop order (plane, tap, slice), 2 KiB per step: steps 0 .. 6 of a phase load at wp_cur + 2 KiB * (tap + 2),
  // steps 7, 8 the first two steps of the NEXT phase (wp_next: next plane, next convolution or next tile) -- both set per phase.
  const char *wp_cur, *wp_next;
  int wblock;                 // bytes of one channel block's fragments of one convolution (2 KiB * 9 * planes)
  // input pieces of the phase: this wave issues piece m (LDS KiB wave + 4 m of the target slot) in step m / PPS; pieces from
  // nreal on are zero fills of the spare KiB (a phase that streams nothing: nreal = 0)
  int nreal, px_soff, px_dst; // px_dst: LDS byte offset of this wave's piece 0
  int off_s;
};

// fp16 rounding of an epilogue value, opaque to mul + cvt fusion like m355_to_half, but not volatile: the sixteen values of a
// pixel block are independent chains the scheduler may interleave (the volatile form cost 210 s_nops per transition)
__device__ __forceinline__ half_t to_half_rn(float v) {
  asm("" : "+v"(v));
  return (half_t)v;
}

// One K step: tap TAP of the current phase.  Entering: st.wa[TAP % 3] holds (or is about to receive) this step's two weight
// fragments, st.wa[(TAP + 1) % 3] the next step's (in flight), fragment set 0 of fb holds K slice 0 of this step.
// Straight-line code (no branch: the input pieces of a phase that streams nothing go to the spare KiB with out-of-range
// offsets, i.e. as zero fills), so that the MFMAs, the LDS reads and the vector-memory issues of a step are ONE scheduling
// region and the sched_group_barriers below can pin  MFMA, ds_read, (vmem)  triples: with one wave per SIMD every
// instruction that is not issued beside an executing MFMA is matrix-pipe idle time.
template <class C, int NPBC, int TAP>
__device__ __forceinline__ void planes_step(PState<C>& st, int wc) {
  constexpr int NT = (TAP + 1) % 9, CUR = TAP % 3, LD = (TAP + 2) % 3;
  constexpr int NPC = TAP < 6 ? C::PPS : 0;             // input pieces of this step
  static_assert(NPBC >= 3 + NPC, "pixel blocks per wave");
  // weight fragments of step g + 2 (always issued: past the block's last step the pointers keep walking valid memory)
  const char* const wsrc = (TAP < 7 ? st.wp_cur + (TAP + 2) * 2048 : st.wp_next + (TAP - 7) * 2048) + st.lane16;
  // K slice 0: MFMAs interleaved with the reads of slice 1, the two weight loads and the input pieces
  const int b1 = st.pcur + (st.tb[TAP] ^ 32);
#pragma unroll
  for (int k = 0; k < NPBC; ++k) {
    st.acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(st.wa[CUR][0], st.fb[0][k], st.acc[k], 0, 0, 0);
    st.fb[1][k] = *(const half8*)(st.smem + b1 + k * (2048 * C::WP));
    if (k == 0) st.wa[LD][0] = *(const half8*)wsrc;
    if (k == 1) st.wa[LD][1] = *(const half8*)(wsrc + 1024);
    if (k >= 2 && k < 2 + NPC) {
      const int m = TAP * C::PPS + (k - 2);
      const bool real = m < st.nreal;
      dma16(st.rs_x, real ? st.pvoff[m] : (int)0x80000000, st.px_soff, st.smem + (real ? st.px_dst + m * 4096 : st.off_s));
    }
  }
#pragma unroll
  for (int k = 0; k < NPBC; ++k) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // 1 DS read
    if (k < 2 + NPC) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // 1 VMEM read
  }
  __builtin_amdgcn_sched_barrier(0);
  if (TAP == 8) {
    // phase end: this wave's pieces of the next plane have landed (issued in steps 0 .. 5; younger: the six weight loads of steps
    // 6, 7, 8), then every wave's
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  }
  // K slice 1: MFMAs interleaved with the reads of the next step's slice 0
  const int b0 = (TAP < 8 ? st.pcur : st.pnext) + st.tb[NT];
#pragma unroll
  for (int k = 0; k < NPBC; ++k) {
    st.acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(st.wa[CUR][1], st.fb[1][k], st.acc[k], 0, 0, 0);
    st.fb[0][k] = *(const half8*)(st.smem + b0 + k * (2048 * C::WP));
  }
#pragma unroll
  for (int k = 0; k < NPBC; ++k) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 1);
  }
  __builtin_amdgcn_sched_barrier(0);
}

// SiLU of eight values, the five instructions of m355_silu (x * rcp(1 + exp2(-log2e * x)): the same bits) issued as five rows
// of eight independent instructions.  The compiler schedules the 16 chains of a pixel block one after the other through one
// temporary (register pressure heuristics at 256 VGPRs): every instruction then waits for its predecessor's result and each
// transcendental needs an s_nop before its use -- 54 cycles per value measured; interleaved it is the issue cost, 28.
__device__ __forceinline__ void silu8(float (&v)[8]) {
  float t[8];
  asm("v_mul_f32 %8, 0xbfb8aa3b, %0\n\tv_mul_f32 %9, 0xbfb8aa3b, %1\n\tv_mul_f32 %10, 0xbfb8aa3b, %2\n\tv_mul_f32 %11, 0xbfb8aa3b, %3\n\t"
      "v_mul_f32 %12, 0xbfb8aa3b, %4\n\tv_mul_f32 %13, 0xbfb8aa3b, %5\n\tv_mul_f32 %14, 0xbfb8aa3b, %6\n\tv_mul_f32 %15, 0xbfb8aa3b, %7\n\t"
      "v_exp_f32 %8, %8\n\tv_exp_f32 %9, %9\n\tv_exp_f32 %10, %10\n\tv_exp_f32 %11, %11\n\t"
      "v_exp_f32 %12, %12\n\tv_exp_f32 %13, %13\n\tv_exp_f32 %14, %14\n\tv_exp_f32 %15, %15\n\t"
      "v_add_f32 %8, 1.0, %8\n\tv_add_f32 %9, 1.0, %9\n\tv_add_f32 %10, 1.0, %10\n\tv_add_f32 %11, 1.0, %11\n\t"
      "v_add_f32 %12, 1.0, %12\n\tv_add_f32 %13, 1.0, %13\n\tv_add_f32 %14, 1.0, %14\n\tv_add_f32 %15, 1.0, %15\n\t"
      "v_rcp_f32 %8, %8\n\tv_rcp_f32 %9, %9\n\tv_rcp_f32 %10, %10\n\tv_rcp_f32 %11, %11\n\t"
      "v_rcp_f32 %12, %12\n\tv_rcp_f32 %13, %13\n\tv_rcp_f32 %14, %14\n\tv_rcp_f32 %15, %15\n\t"
      "v_mul_f32 %0, %0, %8\n\tv_mul_f32 %1, %1, %9\n\tv_mul_f32 %2, %2, %10\n\tv_mul_f32 %3, %3, %11\n\t"
      "v_mul_f32 %4, %4, %12\n\tv_mul_f32 %5, %5, %13\n\tv_mul_f32 %6, %6, %14\n\tv_mul_f32 %7, %7, %15"
      : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]),
        "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3]), "=&v"(t[4]), "=&v"(t[5]), "=&v"(t[6]), "=&v"(t[7]));
}

template <class C, int NPBC>
__device__ __forceinline__ void planes_phase(PState<C>& st, int wc) {
  planes_step<C, NPBC, 0>(st, wc);
  planes_step<C, NPBC, 1>(st, wc);
  planes_step<C, NPBC, 2>(st, wc);
  planes_step<C, NPBC, 3>(st, wc);
  planes_step<C, NPBC, 4>(st, wc);
  planes_step<C, NPBC, 5>(st, wc);
  planes_step<C, NPBC, 6>(st, wc);
  planes_step<C, NPBC, 7>(st, wc);
  planes_step<C, NPBC, 8>(st, wc);
}

struct PTile {
  int b, y0, ch;   // image, first output row, first output channel
};

template <class C>
__global__ __launch_bounds__(256, 1) void planes_kernel(const PlanesArgs a, const PlanesGeom g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;
  const int lrow = lane >> 2, lchunk = lane & 3;
  const int wc = wave % C::WC, wp = wave / C::WC;
  const int H = a.H, W = a.W, PW = g.PW, R = g.R;
  const int nwg = gridDim.x, ntiles = g.ntiles;
  unsigned long long stamp[6] = {0, 0, 0, 0, 0, 0};
  if (a.stamps) stamp[0] = __builtin_amdgcn_s_memtime();

  // XCD-aware persistent walk (as conv3x3_m32.hip): the virtual blocks of one XCD cover a contiguous run of tiles; channel
  // tiles fastest (they share the input slab in L2), then slabs, then images.
  auto decode = [&](int vb) __attribute__((always_inline)) {
    const int xcd = vb & 7, q = ntiles >> 3, r = ntiles & 7;
    const int L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (vb >> 3);
    PTile t;
    const int tch = L % g.tiles_ch;
    const int rest = L / g.tiles_ch;
    t.b = rest / g.nslab;
    t.y0 = (rest - t.b * g.nslab) * R;
    t.ch = tch * 32 * C::WC;
    return t;
  };

  PState<C> st;
  st.smem = smem;
  st.wave = wave;
  st.lane16 = lane * 16;
  st.off_s = g.off_s;
  st.wblock = 2048 * 9 * g.NP;
  const int nreal_wave = (g.npieces - wave + 3) / 4;     // pieces of a plane this wave issues
  const int img_stride = (int)a.x_bstride * 2;
  st.rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)((a.B - 1) * a.x_bstride + (long)H * W * a.ldx) * 2, 0x00020000);

  // input pieces of a tile: piece k = wave + 4 m covers LDS rows 16 k .. 16 k + 15 = storage indices; index j = rj * PW + cj is
  // image pixel (y0 - HALO + rj, cj - 1); outside the image (and the shared zero column cj = 0) the offset fails the range
  // check and the LDS-DMA writes zeros
  constexpr int HALO = C::PAIR ? 2 : 1;
  const int xrows_valid = (R + 2 * HALO) * PW;
  auto piece_offsets = [&](const PTile& t) __attribute__((always_inline)) {
#pragma unroll
    for (int m = 0; m < C::PIT; ++m) {
      const int j = 16 * (wave + 4 * m) + lrow;
      const int rj = j / PW, cj = j - rj * PW;
      const int iy = t.y0 - HALO + rj, ix = cj - 1;
      const bool ok = j < xrows_valid && cj >= 1 && (unsigned)iy < (unsigned)H;
      const int lc = lchunk ^ ((j >> 2) & 3);
      st.pvoff[m] = ok ? ((iy * W + ix) * a.ldx + lc * 8) * 2 : (int)0x80000000;
    }
  };
  // B fragment offsets of the nine taps: storage index of this lane's pixel of block 0 + the tap's shift
  const int idx0 = 32 * wp + l31;
  auto tap_offsets = [&](int dshift) __attribute__((always_inline)) {
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int row = idx0 + (t / 3) * PW + (t % 3) + dshift;
      st.tb[t] = row * ROWB + ((h ^ ((row >> 2) & 3)) << 4);
    }
  };
  auto load_bias = [&](const float* bias, int ch0) __attribute__((always_inline)) {
    const float* bp = bias + ch0 + wc * 32 + 16 * h;
    float16v bv;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4v v = *(const float4v*)(bp + q * 4);
      bv[q * 4 + 0] = v[0]; bv[q * 4 + 1] = v[1]; bv[q * 4 + 2] = v[2]; bv[q * 4 + 3] = v[3];
    }
    return bv;
  };

  // ---- the block's stream of tiles
  int vb = blockIdx.x;
  PTile cur = decode(vb), nxt = cur;
  bool more = vb + nwg < ntiles;
  if (more) nxt = decode(vb + nwg);
  const char* const wbase_a = C::PAIR ? (const char*)a.wfa + (long)wc * st.wblock : nullptr;   // pair: this wave's channel block in either conv
  const char* wbase_b = (const char*)a.wfb + (long)(cur.ch / 32 + wc) * st.wblock;               // second / only conv, this tile
  const char* wbase_b_next = (const char*)a.wfb + (long)(nxt.ch / 32 + wc) * st.wblock;          // ... the block's next tile

  // ---- prologue: input plane 0 into slot 0, the weight fragments of steps 0 and 1
  piece_offsets(cur);
#pragma unroll
  for (int m = 0; m < C::PIT; ++m) {
    const int k = wave + 4 * m;
    if (k < g.npieces) dma16(st.rs_x, st.pvoff[m], cur.b * img_stride, smem + g.off_x + k * 1024);
  }
  {
    const char* w0 = (C::PAIR ? wbase_a : wbase_b) + st.lane16;
#pragma unroll
    for (int sgi = 0; sgi < 2; ++sgi) {
      st.wa[sgi][0] = *(const half8*)(w0 + sgi * 2048);
      st.wa[sgi][1] = *(const half8*)(w0 + sgi * 2048 + 1024);
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (a.stamps) stamp[1] = __builtin_amdgcn_s_memtime();

  const int xslot = g.xrows * ROWB;   // bytes of an input ring slot
  const int iplane = g.irows * ROWB;  // bytes of an intermediate plane
  int xg = 0;                         // input planes consumed so far by this block (ring slot = xg & 1)

  while (true) {
    // =========================== first convolution (pair mode): R + 2 rows of the hidden tensor ===========================
    if constexpr (C::PAIR) {
      tap_offsets(-1);
#pragma unroll
      for (int k = 0; k < C::NPB1; ++k)
#pragma unroll
        for (int j = 0; j < 16; ++j) st.acc[k][j] = 0.f;
      // slice 0 of step 0 (plane 0 of the tile)
      {
        const int p0 = g.off_x + (xg & 1) * xslot;
#pragma unroll
        for (int k = 0; k < C::NPB1; ++k) st.fb[0][k] = *(const half8*)(smem + p0 + st.tb[0] + k * (2048 * C::WP));
      }
      for (int p = 0; p < g.NP; ++p) {
        st.pcur = g.off_x + (xg & 1) * xslot;
        st.pnext = p + 1 < g.NP ? g.off_x + ((xg + 1) & 1) * xslot : st.pcur;
        st.nreal = p + 1 < g.NP ? nreal_wave : 0;   // stream plane p + 1 of this tile into the other slot
        st.px_soff = cur.b * img_stride + (p + 1) * 64;
        st.px_dst = g.off_x + ((xg + 1) & 1) * xslot + wave * 1024;
        st.wp_cur = wbase_a + p * (9 * 2048);
        st.wp_next = p + 1 < g.NP ? st.wp_cur + 9 * 2048 : wbase_b;
        planes_phase<C, C::NPB1>(st, wc);
        ++xg;
      }
      if (a.stamps) stamp[2] = __builtin_amdgcn_s_memtime();
      // ---- transition: bias, SiLU, fp16 into plane wc of the hidden tensor; then the zero padding of the second conv is
      // written OVER it (the shared pad column ci = 0 of every row, the rows outside the image): no per-value select
      if (more) piece_offsets(nxt);   // every piece of this tile has been issued: the loader moves on
      {
        const float16v bv = load_bias(a.ba, 0);
        char* const plane = smem + wc * iplane;
#pragma unroll
        for (int k = 0; k < C::NPB1; ++k) {
          const int i = 32 * (wp + C::WP * k) + l31;
          half8 o[2];
#pragma unroll
          for (int hh = 0; hh < 2; ++hh) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = st.acc[k][hh * 8 + j] + bv[hh * 8 + j];
            silu8(v);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[hh][j] = (half_t)v[j];
          }
          char* rowp = plane + i * ROWB;
          const int sw = (i >> 2) & 3;
          *(half8*)(rowp + (((2 * h) ^ sw) << 4)) = o[0];
          *(half8*)(rowp + (((2 * h + 1) ^ sw) << 4)) = o[1];
          __builtin_amdgcn_sched_barrier(0);   // one pixel block at a time: 16 accumulator values in VGPRs, not 144
        }
        // zero padding (this wave's writes are ordered: LDS operations of one wave complete in order).  Only the wp == 0 wave of a
        // channel block does it -- after a barrier when a second pixel group wrote parts of the plane.
        if (C::WP > 1) {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
        }
        if (wp == 0) {
          const half8 z = {0, 0, 0, 0, 0, 0, 0, 0};
          // pad column: rows ri * PW, ri = 0 .. R + 2 (the last one is the pad behind the last row); 4 lanes x 16 bytes each
          for (int e = lane; e < 4 * (R + 3); e += 64) *(half8*)(plane + (e >> 2) * PW * ROWB + ((e & 3) << 4)) = z;
          // rows outside the image: only the first and the last slab of an image have any
#pragma unroll 1
          for (int ri = 0; ri < R + 2; ++ri) {
            const int iy = cur.y0 - 1 + ri;
            if ((unsigned)iy < (unsigned)H) continue;
            for (int e = lane; e < 4 * PW; e += 64) *(half8*)(plane + (ri * PW + (e >> 2)) * ROWB + ((e & 3) << 4)) = z;
          }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (a.stamps) stamp[3] = __builtin_amdgcn_s_memtime();
    }
    // =========================== second (or only) convolution: R rows ===========================
    {
      tap_offsets(0);
#pragma unroll
      for (int k = 0; k < C::NPB2; ++k)
#pragma unroll
        for (int j = 0; j < 16; ++j) st.acc[k][j] = 0.f;
      {
        const int p0 = C::PAIR ? 0 : g.off_x + (xg & 1) * xslot;
#pragma unroll
        for (int k = 0; k < C::NPB2; ++k) st.fb[0][k] = *(const half8*)(smem + p0 + st.tb[0] + k * (2048 * C::WP));
      }
      const int npl = C::PAIR ? g.NPo : g.NP;
      for (int p = 0; p < npl; ++p) {
        bool pieces = false;
        if (C::PAIR) {
          st.pcur = p * iplane;
          st.pnext = p + 1 < npl ? (p + 1) * iplane : st.pcur;
          if (p == 0 && more) {   // plane 0 of the next tile into the slot the next tile starts with
            pieces = true;
            st.px_soff = nxt.b * img_stride;
            st.px_dst = g.off_x + (xg & 1) * xslot + wave * 1024;
          }
        } else {
          st.pcur = g.off_x + (xg & 1) * xslot;
          st.pnext = g.off_x + ((xg + 1) & 1) * xslot;
          st.px_dst = st.pnext + wave * 1024;
          if (p + 1 < g.NP) {
            pieces = true;
            st.px_soff = cur.b * img_stride + (p + 1) * 64;
          } else if (more) {
            piece_offsets(nxt);
            pieces = true;
            st.px_soff = nxt.b * img_stride;
          } else {
            st.pnext = st.pcur;
          }
          ++xg;
        }
        st.nreal = pieces ? nreal_wave : 0;
        st.wp_cur = wbase_b + p * (9 * 2048);
        st.wp_next = p + 1 < npl ? st.wp_cur + 9 * 2048 : (C::PAIR ? wbase_a : wbase_b_next);
        planes_phase<C, C::NPB2>(st, wc);
      }
      if (a.stamps) stamp[4] = __builtin_amdgcn_s_memtime();
      // ---- epilogue: bias, SiLU, residual, fp16, two 16-byte stores per pixel.  Pixel q -> (row, column) by an exact float
      // division (q < 2^11); every residual load is issued before the first value is touched.
      {
#pragma clang fp contract(off)
        const float rpw = 1.0f / (float)PW;
        const int c0 = cur.ch + wc * 32 + 16 * h;
        const bool ch_ok = c0 + 16 <= a.Cout;
        int off_y[C::NPB2], off_r[C::NPB2];
        half8 rv[C::NPB2][2];
#pragma unroll
        for (int k = 0; k < C::NPB2; ++k) {
          const int q = 32 * (wp + C::WP * k) + l31;
          const int r = (int)(((float)q + 0.5f) * rpw), c = q - r * PW;
          const int yy = cur.y0 + r;
          const bool ok = c < W && r < R && yy < H && ch_ok;
          const int pix = yy * W + c;
          off_y[k] = ok ? pix * a.ldy + c0 : -1;
          off_r[k] = pix * a.ldr + c0;
          if (a.res && ok) {
            const half_t* rp = a.res + (long)cur.b * a.r_bstride + off_r[k];
            rv[k][0] = *(const half8*)rp;
            rv[k][1] = *(const half8*)(rp + 8);
          }
        }
        const float16v bv2 = load_bias(a.bb, cur.ch);
        half_t* const yb = a.y + (long)cur.b * a.y_bstride;
#pragma unroll
        for (int k = 0; k < C::NPB2; ++k) {
          half8 o[2];
#pragma unroll
          for (int hh = 0; hh < 2; ++hh) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = st.acc[k][hh * 8 + j] + bv2[hh * 8 + j];
            silu8(v);
            if (a.res) {
#pragma unroll
              for (int j = 0; j < 8; ++j) v[j] += (float)rv[k][hh][j];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) o[hh][j] = to_half_rn(v[j]);
          }
          if (off_y[k] >= 0) {
            *(half8*)(yb + off_y[k]) = o[0];
            *(half8*)(yb + off_y[k] + 8) = o[1];
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    if (!more) break;
    vb += nwg;
    cur = nxt;
    more = vb + nwg < ntiles;
    if (more) nxt = decode(vb + nwg);
    wbase_b = wbase_b_next;
    wbase_b_next = (const char*)a.wfb + (long)(nxt.ch / 32 + wc) * st.wblock;
    // (the next tile's first phases end in barriers before anything of this tile's LDS state is overwritten: input slots
    // alternate, the hidden planes are rewritten only in the next transition)
  }
  if (a.stamps && lane == 0) {
    stamp[5] = __builtin_amdgcn_s_memtime();
    unsigned long long* o = a.stamps + ((long)blockIdx.x * 4 + wave) * 8;
#pragma unroll
    for (int i = 0; i < 6; ++i) o[i] = stamp[i];
    o[6] = __builtin_amdgcn_s_memrealtime();
  }
}

// ---- host side ----------------------------------------------------------------------------------------------------------
template <class C>
bool planes_geometry(const PlanesArgs& a, PlanesGeom* g) {
  constexpr int HALO = C::PAIR ? 2 : 1;
  const int PW = a.W + 1;
  int R = 32 * C::NB2 / PW;                                      // R * PW <= 32 * NB2
  if (C::PAIR) R = std::min(R, (32 * C::NB1 - 1) / PW - 2);      // (R + 2) * PW + 1 <= 32 * NB1: the pad after the last row is computed (as zero)
  R = std::min(R, a.H);
  const int NP = a.Cin / 32;
  for (; R >= 1; --R) {
    const int npieces = ((R + 2 * HALO) * PW + 1 + 15) / 16;
    if (npieces > 4 * C::PIT) continue;
    const int inter = C::PAIR ? C::WC * 32 * C::NB1 * ROWB : 0;
    if (inter + 2 * npieces * 1024 + 1024 > LDS_MAX) continue;
    break;
  }
  if (R < 1) return false;
  const int nslab = (a.H + R - 1) / R;
  R = (a.H + nslab - 1) / nslab;                                 // equal slabs
  // the kernel always runs its NPB2 (and NPB1) pixel blocks: refuse geometries that leave them mostly empty
  if ((long)R * PW * 10 < (long)32 * C::NB2 * 6) return false;
  g->R = R; g->nslab = nslab; g->PW = PW;
  g->irows = 32 * C::NB1;
  g->npieces = ((R + 2 * HALO) * PW + 1 + 15) / 16;
  g->xrows = 16 * g->npieces;
  g->NP = NP; g->NPo = C::WC;
  g->tiles_ch = C::PAIR ? 1 : (a.Cout + 32 * C::WC - 1) / (32 * C::WC);
  g->ntiles = a.B * nslab * g->tiles_ch;
  g->off_x = C::PAIR ? C::WC * g->irows * ROWB : 0;
  g->off_s = g->off_x + 2 * g->xrows * ROWB;
  return g->off_s + 1024 <= LDS_MAX;
}

bool planes_common_ok(const PlanesArgs& a) {
  if (!a.x || !a.y || !a.wfb || !a.bb || a.B < 1 || a.H < 1 || a.W < 2) return false;
  if (a.Cin % 32 || a.Cin < 64 || a.ldx % 8 || a.ldy % 8 || !a.act) return false;   // (every 3x3 conv of the graphs has SiLU: the kernels apply it unconditionally)
  if (((long)(a.B - 1) * a.x_bstride + (long)a.H * a.W * a.ldx) * 2 >= (1L << 31)) return false;   // one buffer descriptor over the input
  if ((long)a.cblocks_b * (a.Cin / 32) * 18 * 1024 >= (1L << 31)) return false;                        // and one over the fragments
  return true;
}

template <class C>
int planes_launch(const PlanesArgs& a, const PlanesGeom& g, hipStream_t s) {
  auto k = planes_kernel<C>;
  static int slots = 0;
  if (!slots) {
    hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX);
    if (e != hipSuccess) return (int)e;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
      return -2;
    slots = cus & ~7;   // one block per CU; the XCD-aware tile order needs gridDim.x % 8 == 0 whenever a block walks > 1 tile
    if (slots < 8) slots = 8;
  }
  const int grid = g.ntiles <= slots ? g.ntiles : slots;
  hipLaunchKernelGGL(k, dim3(grid), dim3(256), g.off_s + 1024, s, a, g);
  return (int)hipGetLastError();
}

using P128 = PCfg<4, 9, 7, true, 1>;    // 128 hidden channels: 40 x 40 maps in slabs of 5 rows (9 / 7 pixel blocks per wave)
using P64 = PCfg<2, 8, 6, true, 2>;     // 64 hidden channels: 80 x 80 maps in slabs of 4 rows (16 / 12 pixel blocks over two pixel groups)
using S64 = PCfg<2, 0, 4, false, 1>;    // single conv, 64-channel tiles: 20 x 20 maps in slabs of 10 rows

}  // namespace

// Shape-only eligibility (graph construction): an instance exists for C hidden channels and its row slabs fit LDS.
bool bneck_pair_shape_ok(int C, int H, int W) {
  PlanesArgs a{};
  a.H = H; a.W = W; a.B = 1; a.Cin = a.Cout = C;
  PlanesGeom g;
  if (C == 128) return planes_geometry<P128>(a, &g);
  if (C == 64) return planes_geometry<P64>(a, &g);
  return false;
}

// A whole Bottleneck (two 3x3 convs, hidden = in = out channels) in one launch.
bool bneck_pair_ok(const PlanesArgs& a) {
  if (!planes_common_ok(a) || !a.wfa || !a.ba) return false;
  if (a.Cin != a.Cout || a.cblocks_a * 32 < a.Cin || a.cblocks_b * 32 < a.Cin) return false;
  PlanesGeom g;
  if (a.Cin == 128) return planes_geometry<P128>(a, &g);
  if (a.Cin == 64) return planes_geometry<P64>(a, &g);
  return false;
}

int launch_bneck_pair(const PlanesArgs& a, hipStream_t s) {
  if (!bneck_pair_ok(a)) return -1;
  PlanesGeom g;
  if (a.Cin == 128) return planes_geometry<P128>(a, &g) ? planes_launch<P128>(a, g, s) : -1;
  return planes_geometry<P64>(a, &g) ? planes_launch<P64>(a, g, s) : -1;
}

// One 3x3 conv (+ SiLU, + residual) over row slabs, 64-channel output tiles.
bool conv3x3_planes_ok(const PlanesArgs& a) {
  if (!planes_common_ok(a)) return false;
  if ((a.Cin / 32) % 2 || a.Cout % 8 || a.cblocks_b * 32 < (a.Cout + 63) / 64 * 64) return false;   // (the input ring alternates slots across tiles: even plane count)
  PlanesGeom g;
  return planes_geometry<S64>(a, &g);
}

int launch_conv3x3_planes(const PlanesArgs& a, hipStream_t s) {
  if (!conv3x3_planes_ok(a)) return -1;
  PlanesGeom g;
  return planes_geometry<S64>(a, &g) ? planes_launch<S64>(a, g, s) : -1;
}

}  // namespace m355
