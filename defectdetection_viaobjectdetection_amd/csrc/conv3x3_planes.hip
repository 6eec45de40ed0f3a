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
constexpr int NWS = 4;        // weight ring slots
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
  int off_x, off_w, off_s;  // LDS byte offsets: input ring, weight ring, scratch piece
};

template <int WC_, int NPB1_, int NPB2_, bool PAIR_, int PPS_>
struct PCfg {
  static constexpr int WC = WC_, WP = 4 / WC_, NPB1 = NPB1_, NPB2 = NPB2_, PPS = PPS_;
  static constexpr bool PAIR = PAIR_;
  static constexpr int NB1 = WP * NPB1, NB2 = WP * NPB2;
  static constexpr int NPB = NPB1 > NPB2 ? NPB1 : NPB2;
  static constexpr int WSTAGE = 32 * WC * ROWB;   // bytes per weight stage
  static constexpr int W_IT = WC / 2;             // weight DMA instructions per wave and stage
  static constexpr int PIT = 6 * PPS;             // input pieces per wave and plane (issued in steps 0 .. 5)
  static_assert(WC == 2 || WC == 4, "channel blocks per block");
};

template <class C>
struct PState {
  float16v acc[C::NPB];
  half8 fa[2], fb[2][C::NPB];
  int ta;                     // A fragment: LDS byte offset inside a weight stage, K slice 0 (slice 1: ^ 32 -- chunk (2 s + h) ^ swizzle)
  int tb[9];                  // B fragment: LDS byte offset inside a plane for (tap, slice 0), pixel block 0 of this wave
  int pvoff[C::PIT];          // per-lane source offsets of the input pieces of the slab the loader is on
  int wvoff[C::W_IT];
  __amdgpu_buffer_rsrc_t rs_x, rs_wa, rs_wb;
  char* smem;
  int wave;
  // consumer side
  int wslot;                  // ring slot of the current step's weight stage
  int pcur, pnext;            // LDS byte offset of the current / the next phase's plane
  bool last_phase;            // no step follows the phase's last one without a gap (transition / epilogue in between)
  // weight loader cursor: stage (conv, plane, tap) that step g issues = the stage of step g + 3
  int lw_conv, lw_p, lw_t, lw_left, lw_ch0, lw_ch0_next;
  bool drain;                 // a weight issue was skipped (end of the block's stream): counted waits no longer hold
  // input pieces of the phase
  bool pieces;                // this phase streams an input plane
  int px_soff, px_dst;        // scalar source offset (image, plane) and LDS byte offset of the target slot
  int cin2, kpa, kpb;         // bytes per tap in a weight row; bytes per weight row (first / second conv)
  int NP;
  int off_w, off_s, npieces;
};

// s_waitcnt vmcnt(n) lgkmcnt(0) with a compile-time n (n < 64)
template <int N>
__device__ __forceinline__ void wait_vm_lgkm0() {
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory");
}

// One K step: tap TAP of the current phase.  Entering, fragment set 0 holds K slice 0 of this step.
//   NPBC = pixel blocks of this convolution.
template <class C, int NPBC, int TAP>
__device__ __forceinline__ void planes_step(PState<C>& st) {
  constexpr int NT = (TAP + 1) % 9;
  char* const wb = st.smem + st.off_w + st.wslot * C::WSTAGE;
  // K slice 1 of this step
  st.fa[1] = *(const half8*)(wb + (st.ta ^ 32));
#pragma unroll
  for (int k = 0; k < NPBC; ++k) st.fb[1][k] = *(const half8*)(st.smem + st.pcur + (st.tb[TAP] ^ 32) + k * (2048 * C::WP));
#pragma unroll
  for (int k = 0; k < NPBC; ++k) st.acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(st.fa[0], st.fb[0][k], st.acc[k], 0, 0, 0);
  __builtin_amdgcn_sched_barrier(0);
  // this wave's part of the NEXT step's stage has landed (issued two steps ago; younger: the pieces of that step, the
  // weights and the pieces of the previous step), its own LDS reads have returned
  {
    constexpr int P2 = (TAP >= 2 && TAP - 2 < 6) ? C::PPS : 0;
    constexpr int P1 = (TAP >= 1 && TAP - 1 < 6) ? C::PPS : 0;
    if (st.drain) wait_vm_lgkm0<0>();
    else if (st.pieces) wait_vm_lgkm0<P2 + C::W_IT + P1>();
    else wait_vm_lgkm0<C::W_IT>();
  }
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  const int nslot = (st.wslot + 1) & (NWS - 1);
  // K slice 0 of the next step
  if (TAP < 8 || !st.last_phase) {
    char* const wn = st.smem + st.off_w + nslot * C::WSTAGE;
    const int pn = TAP < 8 ? st.pcur : st.pnext;
    st.fa[0] = *(const half8*)(wn + st.ta);
#pragma unroll
    for (int k = 0; k < NPBC; ++k) st.fb[0][k] = *(const half8*)(st.smem + pn + st.tb[NT] + k * (2048 * C::WP));
  }
  // weights of step g + 3 into the slot of step g - 1
  if (st.lw_left > 0) {
    char* dst = st.smem + st.off_w + ((st.wslot + 3) & (NWS - 1)) * C::WSTAGE + st.wave * (C::W_IT * 1024);
    const int soff = st.lw_ch0 + st.lw_t * st.cin2 + st.lw_p * 64;
    if (C::PAIR && st.lw_conv == 0) {
#pragma unroll
      for (int i = 0; i < C::W_IT; ++i) dma16(st.rs_wa, st.wvoff[i], soff, dst + i * 1024);
    } else {
#pragma unroll
      for (int i = 0; i < C::W_IT; ++i) dma16(st.rs_wb, st.wvoff[i], soff, dst + i * 1024);
    }
    --st.lw_left;
    if (++st.lw_t == 9) {
      st.lw_t = 0;
      if (++st.lw_p == st.NP) {   // next convolution; past a tile's last one: the next tile's channel block
        st.lw_p = 0;
        if (C::PAIR) st.lw_conv ^= 1;
        if (!C::PAIR || st.lw_conv == 0) st.lw_ch0 = st.lw_ch0_next;
      }
    }
  } else {
    st.drain = true;
  }
  // input pieces of the phase's target plane: PPS per step in steps 0 .. 5 (a piece past the plane goes to the scratch KiB
  // with an out-of-range offset, so that the number of vector-memory operations per step is a compile-time constant)
  if (TAP < 6 && st.pieces) {
#pragma unroll
    for (int i = 0; i < C::PPS; ++i) {
      const int m = TAP * C::PPS + i;
      const int k = st.wave + 4 * m;
      const bool real = k < st.npieces;
      dma16(st.rs_x, real ? st.pvoff[m] : (int)0x80000000, st.px_soff, st.smem + (real ? st.px_dst + k * 1024 : st.off_s));
    }
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int k = 0; k < NPBC; ++k) st.acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(st.fa[1], st.fb[1][k], st.acc[k], 0, 0, 0);
  st.wslot = nslot;
}

template <class C, int NPBC>
__device__ __forceinline__ void planes_phase(PState<C>& st) {
  planes_step<C, NPBC, 0>(st);
  planes_step<C, NPBC, 1>(st);
  planes_step<C, NPBC, 2>(st);
  planes_step<C, NPBC, 3>(st);
  planes_step<C, NPBC, 4>(st);
  planes_step<C, NPBC, 5>(st);
  planes_step<C, NPBC, 6>(st);
  planes_step<C, NPBC, 7>(st);
  planes_step<C, NPBC, 8>(st);
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
  st.NP = g.NP;
  st.off_w = g.off_w;
  st.off_s = g.off_s;
  st.npieces = g.npieces;
  st.cin2 = a.Cin * 2;
  st.kpa = a.kpad_a * 2;
  st.kpb = a.kpad_b * 2;
  st.drain = false;
  const int img_stride = (int)a.x_bstride * 2;
  st.rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)((a.B - 1) * a.x_bstride + (long)H * W * a.ldx) * 2, 0x00020000);
  st.rs_wb = __builtin_amdgcn_make_buffer_rsrc((void*)a.wb, 0, a.rows_b * a.kpad_b * 2, 0x00020000);
  st.rs_wa = C::PAIR ? __builtin_amdgcn_make_buffer_rsrc((void*)a.wa, 0, a.rows_a * a.kpad_a * 2, 0x00020000) : st.rs_wb;

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
  // weight stage rows: piece k = wave * W_IT + i covers stage rows 16 k .. 16 k + 15; MFMA row 8 q + 4 h + i of a 32-row
  // block holds channel 16 h + 4 q + i, so that accumulator register r of lane-half h is channel 16 h + r
#pragma unroll
  for (int i = 0; i < C::W_IT; ++i) {
    const int r = 16 * (wave * C::W_IT + i) + lrow;
    const int rho = r & 31;
    const int chl = (r & ~31) + 16 * ((rho >> 2) & 1) + 4 * (rho >> 3) + (rho & 3);
    const int lc = lchunk ^ ((r >> 2) & 3);
    st.wvoff[i] = chl * st.kpb + lc * 16;   // both convs of a pair have the same row pitch (same Cin): one offset serves both
  }
  {
    const int r = wc * 32 + l31;
    st.ta = r * ROWB + ((h ^ ((r >> 2) & 3)) << 4);
  }
  // B fragment offsets of the nine taps: storage index of this lane's pixel of block 0 + the tap's shift (D)
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
  const int stages_per_tile = (C::PAIR ? 2 : 1) * 9 * g.NP;
  {
    int mine = 0;
    for (int v = vb; v < ntiles; v += nwg) ++mine;
    st.lw_left = mine * stages_per_tile;
  }
  st.lw_conv = 0; st.lw_p = 0; st.lw_t = 0;
  st.lw_ch0 = cur.ch * st.kpb;
  st.lw_ch0_next = nxt.ch * st.kpb;

  // ---- prologue: input plane 0 into slot 0, weight stages 0, 1, 2
  piece_offsets(cur);
#pragma unroll
  for (int m = 0; m < C::PIT; ++m) {
    const int k = wave + 4 * m;
    if (k < g.npieces) dma16(st.rs_x, st.pvoff[m], cur.b * img_stride, smem + g.off_x + k * 1024);
  }
  st.wslot = 0;
  for (int sgi = 0; sgi < 3; ++sgi) {
    char* dst = smem + g.off_w + sgi * C::WSTAGE + wave * (C::W_IT * 1024);
    const int soff = st.lw_ch0 + st.lw_t * st.cin2 + st.lw_p * 64;
#pragma unroll
    for (int i = 0; i < C::W_IT; ++i) dma16(C::PAIR ? st.rs_wa : st.rs_wb, st.wvoff[i], soff, dst + i * 1024);
    --st.lw_left;
    if (++st.lw_t == 9) {
      st.lw_t = 0;
      if (++st.lw_p == st.NP) { st.lw_p = 0; st.lw_conv ^= 1; }
    }
  }
  // the tail of the intermediate planes is never written by the transition: nothing to clear (32 * NB1 > (R + 2) * PW)
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (a.stamps) stamp[1] = __builtin_amdgcn_s_memtime();

  const int xslot = g.xrows * ROWB;   // bytes of an input ring slot
  const int iplane = g.irows * ROWB;  // bytes of an intermediate plane
  int xg = 0;                         // input planes consumed so far by this block (ring slot = xg & 1)

  while (true) {
    // =========================== first convolution (pair mode): R + 2 rows of the hidden tensor ===========================
    if (C::PAIR) {
      tap_offsets(-1);
#pragma unroll
      for (int k = 0; k < C::NPB1; ++k)
#pragma unroll
        for (int j = 0; j < 16; ++j) st.acc[k][j] = 0.f;
      // slice 0 of step 0 (plane 0 of the tile, weight stage in st.wslot)
      {
        const int p0 = g.off_x + (xg & 1) * xslot;
        st.fa[0] = *(const half8*)(smem + g.off_w + st.wslot * C::WSTAGE + st.ta);
#pragma unroll
        for (int k = 0; k < C::NPB1; ++k) st.fb[0][k] = *(const half8*)(smem + p0 + st.tb[0] + k * (2048 * C::WP));
      }
      for (int p = 0; p < g.NP; ++p) {
        st.pcur = g.off_x + (xg & 1) * xslot;
        st.pnext = g.off_x + ((xg + 1) & 1) * xslot;
        st.last_phase = p == g.NP - 1;
        st.pieces = p + 1 < g.NP;   // stream plane p + 1 of this tile into the other slot
        st.px_soff = cur.b * img_stride + (p + 1) * 64;
        st.px_dst = st.pnext;
        planes_phase<C, C::NPB1>(st);
        ++xg;
      }
      if (a.stamps) stamp[2] = __builtin_amdgcn_s_memtime();
      // ---- transition: bias is in the accumulators; SiLU, zero outside the image (the zero padding of the second conv), fp16,
      // into plane wc of the hidden tensor
      if (more) piece_offsets(nxt);   // every piece of this tile has been issued: the loader moves on
      {
        const float16v bv = load_bias(a.ba, 0);
#pragma unroll
        for (int k = 0; k < C::NPB1; ++k) {
          const int i = 32 * (wp + C::WP * k) + l31;
          const int ri = i / PW, ci = i - ri * PW;
          const int iy = cur.y0 - 1 + ri;
          const bool ok = ci >= 1 && ri < R + 2 && (unsigned)iy < (unsigned)H;
          half8 o[2];
#pragma unroll
          for (int hh = 0; hh < 2; ++hh)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              float v = st.acc[k][hh * 8 + j] + bv[hh * 8 + j];
              if (a.act) v = m355_silu(v);
              o[hh][j] = m355_to_half(ok ? v : 0.f);
            }
          char* rowp = smem + wc * iplane + i * ROWB;
          const int sw = (i >> 2) & 3;
          *(half8*)(rowp + (((2 * h) ^ sw) << 4)) = o[0];
          *(half8*)(rowp + (((2 * h + 1) ^ sw) << 4)) = o[1];
          __builtin_amdgcn_sched_barrier(0);   // one pixel block at a time: 16 accumulator values in VGPRs, not 144
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
        st.fa[0] = *(const half8*)(smem + g.off_w + st.wslot * C::WSTAGE + st.ta);
#pragma unroll
        for (int k = 0; k < C::NPB2; ++k) st.fb[0][k] = *(const half8*)(smem + p0 + st.tb[0] + k * (2048 * C::WP));
      }
      const int npl = C::PAIR ? g.NPo : g.NP;
      for (int p = 0; p < npl; ++p) {
        st.last_phase = p == npl - 1;
        bool pieces = false;
        if (C::PAIR) {
          st.pcur = p * iplane;
          st.pnext = (p + 1) * iplane;
          if (p == 0 && more) {   // plane 0 of the next tile into the slot the next tile starts with
            pieces = true;
            st.px_soff = nxt.b * img_stride;
            st.px_dst = g.off_x + (xg & 1) * xslot;
          }
        } else {
          st.pcur = g.off_x + (xg & 1) * xslot;
          st.pnext = g.off_x + ((xg + 1) & 1) * xslot;
          if (p + 1 < g.NP) {
            pieces = true;
            st.px_soff = cur.b * img_stride + (p + 1) * 64;
            st.px_dst = st.pnext;
          } else if (more) {
            piece_offsets(nxt);
            pieces = true;
            st.px_soff = nxt.b * img_stride;
            st.px_dst = st.pnext;
          }
          ++xg;
        }
        st.pieces = pieces;
        planes_phase<C, C::NPB2>(st);
      }
      if (a.stamps) stamp[4] = __builtin_amdgcn_s_memtime();
      // ---- epilogue: bias, SiLU, residual, fp16, two 16-byte stores per pixel
      const float16v bv2 = load_bias(a.bb, cur.ch);
#pragma unroll
      for (int k = 0; k < C::NPB2; ++k) {
#pragma clang fp contract(off)
        const int q = 32 * (wp + C::WP * k) + l31;
        const int r = q / PW, c = q - r * PW;
        const int yy = cur.y0 + r;
        const bool ok = c < W && r < R && yy < H;
        const long pix = (long)yy * W + c;
        const int c0 = cur.ch + wc * 32 + 16 * h;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          if (!ok || c0 + hh * 8 + 8 > a.Cout) continue;
          float v[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = st.acc[k][hh * 8 + j] + bv2[hh * 8 + j];
          if (a.act) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = m355_silu(v[j]);
          }
          if (a.res) {
            const half8 rv = *(const half8*)(a.res + (long)cur.b * a.r_bstride + pix * a.ldr + c0 + hh * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += (float)rv[j];
          }
          half8 o;
#pragma unroll
          for (int j = 0; j < 8; ++j) o[j] = m355_to_half(v[j]);
          *(half8*)(a.y + (long)cur.b * a.y_bstride + pix * a.ldy + c0 + hh * 8) = o;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (!more) break;
    vb += nwg;
    cur = nxt;
    more = vb + nwg < ntiles;
    if (more) nxt = decode(vb + nwg);
    st.lw_ch0_next = nxt.ch * st.kpb;
    // every wave has left the previous tile's hidden planes / last input plane before the next tile overwrites them: the
    // barriers of the next tile's first steps order that (the transition is > NP phases away; input slots alternate)
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
    if (inter + 2 * npieces * 1024 + NWS * C::WSTAGE + 1024 > LDS_MAX) continue;
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
  g->off_w = g->off_x + 2 * g->xrows * ROWB;
  g->off_s = g->off_w + NWS * C::WSTAGE;
  return g->off_s + 1024 <= LDS_MAX;
}

bool planes_common_ok(const PlanesArgs& a) {
  if (!a.x || !a.y || !a.wb || !a.bb || a.B < 1 || a.H < 1 || a.W < 2) return false;
  if (a.Cin % 32 || a.Cin < 64 || a.ldx % 8 || a.ldy % 8 || a.kpad_b % 8 || a.kpad_b < 9 * a.Cin) return false;
  if (((long)(a.B - 1) * a.x_bstride + (long)a.H * a.W * a.ldx) * 2 >= (1L << 31)) return false;   // one buffer descriptor over the input
  if ((long)a.rows_b * a.kpad_b * 2 >= (1L << 31)) return false;
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

// A whole Bottleneck (two 3x3 convs, hidden = in = out channels) in one launch.
bool bneck_pair_ok(const PlanesArgs& a) {
  if (!planes_common_ok(a) || !a.wa || !a.ba) return false;
  if (a.Cin != a.Cout || a.kpad_a != a.kpad_b || a.rows_a < a.Cin || a.rows_b < a.Cin) return false;
  if ((long)a.rows_a * a.kpad_a * 2 >= (1L << 31)) return false;
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
  if ((a.Cin / 32) % 2 || a.Cout % 8 || a.rows_b < (a.Cout + 63) / 64 * 64) return false;   // (the input ring alternates slots across tiles: even plane count)
  PlanesGeom g;
  return planes_geometry<S64>(a, &g);
}

int launch_conv3x3_planes(const PlanesArgs& a, hipStream_t s) {
  if (!conv3x3_planes_ok(a)) return -1;
  PlanesGeom g;
  return planes_geometry<S64>(a, &g) ? planes_launch<S64>(a, g, s) : -1;
}

}  // namespace m355
