// 3x3 / stride-1 / pad-1 NHWC fp16 convolution on v_mfma_f32_32x32x16_f16 with an LDS-staged halo patch (gfx950).
//
// Replaces: the Conv+BN+SiLU 3x3 layers of the C2f bottlenecks and of the Detect/Segment head
// that upstream reaches through torch.nn.functional.conv2d (call site: BscanBased/yolo8_seg_predict.py:8).
//
// Why a third 3x3 kernel.  The 16x16x32 halo kernels (conv3x3_halo.hip, conv3x3_wide.hip) are ISSUE-bound: a SIMD
// issues about one instruction per four cycles, a 16x16x32 MFMA holds the issue port for 8 of its 16 cycles, so at
// most two other instructions per MFMA hide behind the matrix pipe and those kernels spend 5-10.
// The 32x32x16 form does twice the work per instruction (32 cycles, 8 of them on the issue port): six free slots per
// MFMA.  This kernel is built so that the per-step instruction stream is close to the minimum the data flow needs:
//   * every LDS fragment address is  (one of 16 per-lane base registers) + (compile-time immediate): the nine taps and
//     both patch buffers are unrolled, the tap's row shift kh and the column block are additive in the immediate, the
//     tap's column shift kw selects one of three base sets (the XOR swizzle depends on the patch column only);
//   * every global->LDS transfer is ONE `buffer_load_dwordx4 ... offen lds` whose per-lane offset register changes only
//     from tile to tile: the K position of a step travels in the scalar offset, image borders are out-of-range
//     offsets -- measured on MI355X (tools/probes/lds_dma_oob_probe.hip): a lane that fails the buffer range check
//     WRITES ZEROS to its LDS slot -- so there is no EXEC masking, no zero page, no pre-zeroing and no 64-bit address
//     arithmetic in the loop;
//   * the accumulators start at the bias.
//
// GEMM orientation as in the other conv kernels: D[channel][pixel] += W[channel][k] * patch[pixel + tap][k].
//   v_mfma_f32_32x32x16_f16: lane l holds A[row l&31][k 8(l>>5)..+7], B[k 8(l>>5)..+7][col l&31],
//   D[row (r&3) + 8(r>>2) + 4(l>>5)][col l&31] in register r of 16.
// A column block = 2 tile rows x 16 pixels (lanes 0-15 / 16-31).  Weight rows are permuted on the DMA source side so
// that lane-half h owns 16 CONSECUTIVE channels per 32-row block: two 16-byte NHWC stores per pixel and block.
//
// LDS image: one 128-byte row per patch pixel (64 input channels) / per weight row, the 16-byte chunk index XOR-ed with
// f = (patch column >> 1) & 7 (weights: (row >> 1) & 7), applied to the DMA source address and to the ds_read: every
// ds_read_b128 of a 32-pixel x 8-half fragment is bank-conflict free for all nine tap shifts (its four 16-lane service
// groups each see sixteen distinct patch columns: eight even, eight odd, f distinct within each parity).
//
// Block = 4 waves, 2 blocks per CU; tile = BCH channels x TH rows x 16 pixels, wave = 64 channels x NB row pairs.
// K loop: step = (64-channel chunk, tap) = 4 MFMA slices of K = 16; weights ring of two 64-deep stages; ONE barrier per
// step, before the last slice: by then every wave has read the stage, so the DMA of step n+2 is issued into it and the
// first fragments of step n+1 are read under the last slice's MFMAs.  The patch of the next chunk streams into the
// second patch buffer one piece per step.
//
// PERSISTENT blocks: measured with in-kernel stamps (tools/stamps_m32.py) a one-tile-per-block launch spent 20 % of a
// block's life in the prologue (setup + the first 55 KB of DMA with nothing to compute) and 13 % in the epilogue, so the
// matrix pipe was busy 55 % of the time although the main loop alone keeps it 82-87 % busy.  Here a block walks its tiles
// (virtual block vb = blockIdx.x + k * gridDim.x) and the (tile, chunk) sequence is ONE continuous stream: the loader
// state switches to the next tile at the start of a tile's last chunk, so the next tile's first patch and its first two
// weight stages land under the last chunk's MFMAs and its first fragments are read before the epilogue starts; the two
// blocks of a CU drift apart, one block's epilogue (VALU) runs beside the other's main loop (MFMA).
#include <stdlib.h>

#include "common.h"

namespace m355 {
namespace {

typedef float float16v __attribute__((ext_vector_type(16)));

constexpr int TS = 16;      // tile width in pixels
constexpr int PP = 18;      // patch pitch (pixels): EVEN, so that a pixel's LDS-row parity is its column parity
constexpr int ROWB = 128;   // bytes per LDS row

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, int voff, int soff, char* lds) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds, 16, voff, soff, 0, 0);
}

template <int BCH, int NB>
struct M32 {
  static constexpr int WCH = BCH / 64;            // waves along channels
  static constexpr int WPX = 4 / WCH;             // waves along rows
  static constexpr int TH = 2 * NB * WPX;         // tile rows
  static constexpr int PH = TH + 2;
  static constexpr int PGROUPS = (PH * PP + 7) / 8;
  static constexpr int PATCH_BYTES = PGROUPS * 8 * ROWB;
  static constexpr int P_IT = (PGROUPS + 3) / 4;  // patch DMA instructions per wave and chunk
  static constexpr int WBUF = BCH * ROWB;
  static constexpr int W_IT = BCH / 32;           // weight DMA instructions per wave and step
  static_assert(BCH == 64 || BCH == 128, "channel tile");
  static_assert((NB * 2 + 2) * PP * ROWB + PATCH_BYTES < 65536, "ds_read immediate range");
};

// Everything a step needs, in registers.
template <int BCH, int NB>
struct M32State {
  using C = M32<BCH, NB>;
  float16v acc[2][NB];
  half8 fa[2][2], fb[2][NB];       // fragment double buffer: [set][mt] / [set][nb]
  int ta[4];                       // LDS byte offset of the weight fragment of K slice s (stage 0, mt 0)
  int tb[3][4];                    // LDS byte offset of the patch fragment of (kw, K slice s) (buffer 0, column block 0, kh 0)
  int pvoff[C::P_IT];              // per-lane byte offsets of the patch pieces of the NEXT chunk's tile (out of range: border)
  int wvoff[C::W_IT];
  __amdgpu_buffer_rsrc_t rs_w, rs_x;   // the whole packed weight matrix / the whole input tensor (tile and chunk travel in the scalar offset)
  int kcur, knx, soff_xn;          // byte offset of (channel tile, chunk) of this / the next chunk in the weights; of (image, chunk) of the next chunk
  bool has_next;                   // a chunk follows in this block's stream (same tile or the next tile)
  char* smem;
  int wave, woff;                  // woff = npatch * PATCH_BYTES: start of the weight ring
  int cin2;                        // bytes per tap in a packed weight row
  int dbg;                         // timing-only ablations (tools/stamps_m32.py): 1 no patch DMA, 2 no weight DMA in the loop, 4 no barrier
};

// One K step of the chunk in patch buffer PB, tap TAP.  Entering, fragment set 0 holds K slice 0 of this step.
template <int BCH, int NB, int TAP, int PB>
__device__ __forceinline__ void m32_step(M32State<BCH, NB>& st) {
  using C = M32<BCH, NB>;
  constexpr int KH = TAP / 3, KW = TAP % 3;
  constexpr int PARW = (PB + TAP) & 1;                       // weight stage of this step
  constexpr int NTAP = (TAP + 1) % 9, NPB = TAP == 8 ? (PB ^ 1) : PB;
  constexpr int NKH = NTAP / 3, NKW = NTAP % 3, NPARW = PARW ^ 1;
  constexpr int AOFF = PARW * C::WBUF;
  constexpr int BOFF = PB * C::PATCH_BYTES + KH * PP * ROWB;
#define M32_READ(SET, S)                                                                                      \
  {                                                                                                           \
    _Pragma("unroll") for (int mt = 0; mt < 2; ++mt)                                                          \
        st.fa[SET][mt] = *(const half8*)(st.smem + (st.ta[S] + AOFF + mt * 32 * ROWB));                       \
    _Pragma("unroll") for (int nb = 0; nb < NB; ++nb)                                                         \
        st.fb[SET][nb] = *(const half8*)(st.smem + (st.tb[KW][S] + BOFF + nb * 2 * PP * ROWB));               \
  }
#define M32_MFMA(SET)                                                                                         \
  {                                                                                                           \
    _Pragma("unroll") for (int mt = 0; mt < 2; ++mt) _Pragma("unroll") for (int nb = 0; nb < NB; ++nb)        \
        st.acc[mt][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(st.fa[SET][mt], st.fb[SET][nb], st.acc[mt][nb], 0, 0, 0); \
  }
  M32_READ(1, 1)
  M32_MFMA(0)
  __builtin_amdgcn_sched_barrier(0);
  M32_READ(0, 2)
  M32_MFMA(1)
  __builtin_amdgcn_sched_barrier(0);
  M32_READ(1, 3)
  M32_MFMA(0)
  __builtin_amdgcn_sched_barrier(0);
  // every wave: its reads of this stage have returned, its DMA of the next step (and patch pieces) has landed
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  if (!(st.dbg & 4)) __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  // K slice 0 of the next step (tap 0 of the next chunk after tap 8: only if the stream goes on)
  if (TAP < 8 || st.has_next) {
    constexpr int NA = NPARW * C::WBUF, NBO = NPB * C::PATCH_BYTES + NKH * PP * ROWB;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) st.fa[0][mt] = *(const half8*)(st.smem + (st.ta[0] + NA + mt * 32 * ROWB));
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) st.fb[0][nb] = *(const half8*)(st.smem + (st.tb[NKW][0] + NBO + nb * 2 * PP * ROWB));
  }
  // weights of step n + 2 into the stage this step has just finished with
  {
    char* dst = st.smem + st.woff + PARW * C::WBUF + st.wave * 1024;
    if (st.dbg & 2) {
    } else if (TAP <= 6) {
      const int soff = st.kcur + (TAP + 2) * st.cin2;
#pragma unroll
      for (int i = 0; i < C::W_IT; ++i) dma16(st.rs_w, st.wvoff[i], soff, dst + i * 4096);
    } else if (st.has_next) {
      const int soff = st.knx + (TAP - 7) * st.cin2;
#pragma unroll
      for (int i = 0; i < C::W_IT; ++i) dma16(st.rs_w, st.wvoff[i], soff, dst + i * 4096);
    }
  }
  // one (or two) pieces of the next chunk's patch into the other patch buffer; taps 0..7 only, so that the last piece
  // has landed (this wait + this barrier, one step later) before tap 0 of the next chunk reads it
  if (TAP < 8 && st.has_next && !(st.dbg & 1)) {
#pragma unroll
    for (int i = 0; i < C::P_IT; ++i) {
      if ((i % 8) != TAP) continue;
      if (st.wave + 4 * i < C::PGROUPS)
        dma16(st.rs_x, st.pvoff[i], st.soff_xn, st.smem + (PB ^ 1) * C::PATCH_BYTES + (st.wave + 4 * i) * 1024);
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  M32_MFMA(1)
#undef M32_READ
#undef M32_MFMA
}

template <int BCH, int NB, int PB>
__device__ __forceinline__ void m32_chunk(M32State<BCH, NB>& st) {
  m32_step<BCH, NB, 0, PB>(st);
  m32_step<BCH, NB, 1, PB>(st);
  m32_step<BCH, NB, 2, PB>(st);
  m32_step<BCH, NB, 3, PB>(st);
  m32_step<BCH, NB, 4, PB>(st);
  m32_step<BCH, NB, 5, PB>(st);
  m32_step<BCH, NB, 6, PB>(st);
  m32_step<BCH, NB, 7, PB>(st);
  m32_step<BCH, NB, 8, PB>(st);
}

struct M32Tile {
  int b, y0, x0, ch;   // image, first row, first column, first channel
};

template <int BCH, int NB>
__global__ __launch_bounds__(256, 2) void conv3x3_m32_kernel(const ConvArgs a, int tiles_x, int tiles_y, int nchunks,
                                                            int npatch, int ntiles) {
  using C = M32<BCH, NB>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  unsigned long long st0 = 0, st1 = 0, rt0 = 0;   // diagnostic runs only (tools/stamps_m32.py)
  if (a.stamps) {
    st0 = __builtin_amdgcn_s_memtime();
    rt0 = __builtin_amdgcn_s_memrealtime();
  }
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lrow = lane >> 3, slot = lane & 7;
  const int H = a.Hi, W = a.Wi;
  const int tiles_ch = (a.Cout + BCH - 1) / BCH;
  const int nwg = gridDim.x;

  // ---- persistent walk: virtual block vb = blockIdx.x + k * gridDim.x (gridDim.x is a multiple of 8 or equals ntiles, so
  // vb & 7 is this block's XCD group for every k).  XCD-aware order: the virtual blocks of one XCD cover a contiguous run
  // of tiles; channel tiles fastest, then x, y, image.
  auto decode = [&](int vb) __attribute__((always_inline)) {
    const int xcd = vb & 7, q = ntiles >> 3, r = ntiles & 7;
    const int L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (vb >> 3);
    M32Tile t;
    const int tile_ch = L % tiles_ch;
    int rest = L / tiles_ch;
    const int tx = rest % tiles_x;
    rest /= tiles_x;
    const int ty = rest % tiles_y;
    t.b = rest / tiles_y;
    t.ch = tile_ch * BCH;
    t.y0 = ty * C::TH;
    t.x0 = tx * TS;
    return t;
  };

  M32State<BCH, NB> st;
  st.smem = smem;
  st.wave = wave;
  st.woff = npatch * C::PATCH_BYTES;
  st.cin2 = a.Cin * 2;
  st.dbg = a.dbg;
  // ONE descriptor per operand for the whole kernel (the launcher checked that both fit 31-bit byte offsets)
  const int nimg = a.M / (a.Ho * a.Wo);
  st.rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)((nimg - 1) * a.x_bstride + (long)H * W * a.ldx) * 2, 0x00020000);
  st.rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, tiles_ch * BCH * a.Kpad * 2, 0x00020000);
  const int img_stride = (int)a.x_bstride * 2, wt_stride = a.Kpad * 2;   // bytes per image / per weight row
  // patch pieces of a tile: this wave owns LDS row groups j = wave + 4 i; lane = (row 8 j + lane / 8, 16-byte slot lane % 8).
  // A lane outside the image (or past the patch) gets an offset that fails the range check: its LDS slot receives zeros.
  auto patch_offsets = [&](const M32Tile& t) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < C::P_IT; ++i) {
      const int p = 8 * (wave + 4 * i) + lrow;
      const int py = p / PP, px = p - py * PP;
      const int iy = t.y0 - 1 + py, ix = t.x0 - 1 + px;
      const bool ok = p < C::PH * PP && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
      const int cc = slot ^ ((px >> 1) & 7);
      st.pvoff[i] = ok ? ((iy * W + ix) * a.ldx + cc * 8) * 2 : (int)0x80000000;
    }
  };
  // weights: LDS row R of a stage <- logical channel chl(R): MFMA row 8q + 4h + i of a 32-row block holds channel
  // 16h + 4q + i, so that lane-half h's accumulator registers r = 4q + i are channels 16h .. 16h + 15
#pragma unroll
  for (int i = 0; i < C::W_IT; ++i) {
    const int R = (i * 4 + wave) * 8 + lrow;
    const int rho = R & 31;
    const int chl = (R & ~31) + 16 * ((rho >> 2) & 1) + 4 * (rho >> 3) + (rho & 3);
    const int cc = slot ^ ((R >> 1) & 7);
    st.wvoff[i] = (chl * a.Kpad + cc * 8) * 2;
  }

  // ---- fragment base addresses (tile independent)
  const int wch = wave / C::WPX, wpx = wave % C::WPX;
  const int l31 = lane & 31, h = lane >> 5, x15 = lane & 15, r2 = (lane >> 4) & 1;
#pragma unroll
  for (int s = 0; s < 4; ++s) st.ta[s] = st.woff + (wch * 64 + l31) * ROWB + ((((2 * s + h) ^ ((l31 >> 1) & 7))) << 4);
#pragma unroll
  for (int kw = 0; kw < 3; ++kw) {
    const int px = x15 + kw;
    const int p = (wpx * 2 * NB + r2) * PP + px;
#pragma unroll
    for (int s = 0; s < 4; ++s) st.tb[kw][s] = p * ROWB + (((2 * s + h) ^ ((px >> 1) & 7)) << 4);
  }

  // accumulators start at the bias: register r of (mt, lane-half h) is channel ch + wch*64 + mt*32 + 16h + r
  auto load_bias = [&](const M32Tile& t, float16v (&bv)[2]) __attribute__((always_inline)) {
    const float* bp = a.bias + t.ch + wch * 64 + 16 * h;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4v v = *(const float4v*)(bp + mt * 32 + q * 4);
        bv[mt][q * 4 + 0] = v[0]; bv[mt][q * 4 + 1] = v[1]; bv[mt][q * 4 + 2] = v[2]; bv[mt][q * 4 + 3] = v[3];
      }
  };

  // ---- first tile: patch of chunk 0, weights of steps 0 and 1 (taps 0, 1 of chunk 0)
  int vb = blockIdx.x;
  M32Tile cur = decode(vb), nxt = cur;
  patch_offsets(cur);
#pragma unroll
  for (int i = 0; i < C::P_IT; ++i)
    if (wave + 4 * i < C::PGROUPS) dma16(st.rs_x, st.pvoff[i], cur.b * img_stride, smem + (wave + 4 * i) * 1024);
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int i = 0; i < C::W_IT; ++i)
      dma16(st.rs_w, st.wvoff[i], cur.ch * wt_stride + t * st.cin2, smem + st.woff + t * C::WBUF + wave * 1024 + i * 4096);
  {
    float16v bv[2];
    load_bias(cur, bv);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) st.acc[mt][nb] = bv[mt];
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) st.fa[0][mt] = *(const half8*)(smem + (st.ta[0] + mt * 32 * ROWB));
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) st.fb[0][nb] = *(const half8*)(smem + (st.tb[0][0] + nb * 2 * PP * ROWB));
  if (a.stamps) st1 = __builtin_amdgcn_s_memtime();

  int c = 0, ntile_done = 0;
  // Before a chunk: where does the stream go next?  Same tile -> same buffers, K offset + 128 bytes; last chunk of a tile ->
  // the loader state switches to the block's next tile (this tile's patch pieces have all been issued by now).
  auto begin_chunk = [&]() __attribute__((always_inline)) {
    st.kcur = cur.ch * wt_stride + c * 128;
    if (c + 1 < nchunks) {
      st.has_next = true;
      st.soff_xn = cur.b * img_stride + (c + 1) * 128;
      st.knx = st.kcur + 128;
    } else if (vb + nwg < ntiles) {
      nxt = decode(vb + nwg);
      patch_offsets(nxt);
      st.has_next = true;
      st.soff_xn = nxt.b * img_stride;
      st.knx = nxt.ch * wt_stride;
    } else {
      st.has_next = false;
    }
  };
  // After a chunk: the tile's last chunk ends with the epilogue (SiLU, residual, fp16, two 16-byte stores per pixel and
  // 32-channel block); false when the block's stream is finished.
  auto end_chunk = [&]() __attribute__((always_inline)) -> bool {
    if (c + 1 < nchunks) {
      ++c;
      return true;
    }
    const bool more = vb + nwg < ntiles;
    float16v bn[2];
    if (more) load_bias(nxt, bn);   // lands under the epilogue arithmetic
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
#pragma clang fp contract(off)
      const int yy = cur.y0 + wpx * 2 * NB + nb * 2 + r2, xx = cur.x0 + x15;
      const bool pix_ok = yy < H && xx < W;
      const long pix = (long)yy * W + xx;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const int c0 = cur.ch + wch * 64 + mt * 32 + 16 * h;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          if (!pix_ok || c0 + half * 8 + 8 > a.Cout) continue;
          float v[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = st.acc[mt][nb][half * 8 + j];
          if (a.act) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = m355_silu(v[j]);
          }
          if (a.res) {
            const half8 rv = *(const half8*)(a.res + (long)cur.b * a.r_bstride + pix * a.ldr + c0 + half * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += (float)rv[j];
          }
          half8 o;
#pragma unroll
          for (int j = 0; j < 8; ++j) o[j] = m355_to_half(v[j]);
          *(half8*)((half_t*)a.y + (long)cur.b * a.y_bstride + pix * a.ldy + c0 + half * 8) = o;
        }
      }
    }
    ++ntile_done;
    if (!more) return false;
    vb += nwg;
    cur = nxt;
    c = 0;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) st.acc[mt][nb] = bn[mt];
    return true;
  };

  while (true) {
    begin_chunk();
    m32_chunk<BCH, NB, 0>(st);
    if (!end_chunk()) break;
    begin_chunk();
    m32_chunk<BCH, NB, 1>(st);
    if (!end_chunk()) break;
  }
  if (a.stamps && tid == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned long long* o = a.stamps + (long)blockIdx.x * 8;
    o[0] = st0; o[1] = st1; o[2] = 0; o[3] = __builtin_amdgcn_s_memtime(); o[4] = rt0; o[5] = __builtin_amdgcn_s_memrealtime();
    o[6] = (unsigned long long)ntile_done;
  }
}

template <int BCH, int NB>
int launch_m32(const ConvArgs& a, hipStream_t s) {
  using C = M32<BCH, NB>;
  if (!conv_rows_covered(a, BCH)) return -1;
  const int tiles_x = (a.Wi + TS - 1) / TS, tiles_y = (a.Hi + C::TH - 1) / C::TH;
  const int tiles_ch = (a.Cout + BCH - 1) / BCH;
  const int nchunks = a.Cin / 64;
  const int B = a.M / (a.Ho * a.Wo);
  const int ntiles = B * tiles_y * tiles_x * tiles_ch;
  // a persistent block streams chunk after chunk, also across tiles: it needs both patch buffers unless it never has a
  // second chunk at all (one chunk per tile AND one tile per block)
  auto k = conv3x3_m32_kernel<BCH, NB>;
  static int slots[2] = {0, 0};   // resident blocks for (one, two) patch buffers
  constexpr int LDS1 = C::PATCH_BYTES + 2 * C::WBUF, LDS2 = 2 * C::PATCH_BYTES + 2 * C::WBUF;
  if (!slots[0]) {
    hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, LDS2);
    if (e != hipSuccess) return (int)e;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
      return -2;
    for (int i = 0; i < 2; ++i) {
      int per_cu = (160 * 1024) / (i ? LDS2 : LDS1);
      if (per_cu > 2) per_cu = 2;   // two waves per SIMD: the register budget of __launch_bounds__(256, 2)
      if (per_cu < 1) per_cu = 1;
      const char* ev = getenv("M355_M32_SLOTS");
      slots[i] = ev ? atoi(ev) : per_cu * cus;
      if (slots[i] < 8) slots[i] = 8;
      slots[i] &= ~7;               // the XCD-aware tile order needs gridDim.x % 8 == 0 whenever a block walks > 1 tile
    }
  }
  const bool single = nchunks == 1 && ntiles <= slots[0];
  const int grid = single ? ntiles : (ntiles <= slots[1] ? ntiles : slots[1]);
  hipLaunchKernelGGL(k, dim3(grid), dim3(256), single ? LDS1 : LDS2, s, a, tiles_x, tiles_y, nchunks, single ? 1 : 2, ntiles);
  return (int)hipGetLastError();
}

}  // namespace

// Eligibility: the halo rules (3x3 / s1 / p1, fp16 out, Cin % 64 == 0, Cout >= 64, tiling waste <= 30 %), buffers the
// 32-bit buffer offsets can address.
bool conv3x3_m32_ok(const ConvArgs& a) {
  if (!conv3x3_halo_ok(a)) return false;
  const long nimg = a.M / ((long)a.Ho * a.Wo);
  if (((nimg - 1) * a.x_bstride + (long)a.Hi * a.Wi * a.ldx) * 2 >= (1L << 31)) return false;   // one buffer descriptor over the input
  if ((long)conv_cout_pad(a.Cout) * a.Kpad * 2 >= (1L << 31)) return false;                       // and one over the weights
  return a.Cin <= 4096;
}

// which: 0 = by shape, 1 = <128 ch, 8 rows>, 2 = <64 ch, 16 rows>, 3 = <64 ch, 8 rows>
int launch_conv3x3_m32(const ConvArgs& a, int which, hipStream_t s) {
  if (!conv3x3_m32_ok(a)) return -1;
  if (which == 0) {
    if (a.Cout > 64) which = 1;
    else which = (((a.Hi + 15) / 16) * 16 * 10 <= a.Hi * 12) ? 2 : 3;   // 16-row tiles unless they waste > 20 % of the rows
  }
  if (which == 1) return launch_m32<128, 2>(a, s);
  if (which == 2) return launch_m32<64, 2>(a, s);
  if (which == 3) return launch_m32<64, 1>(a, s);
  return -1;
}

}  // namespace m355
