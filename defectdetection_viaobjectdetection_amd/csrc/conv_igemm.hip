// Implicit-GEMM NHWC fp16 convolution on CDNA4 MFMA (gfx950), fp32 accumulate.
//
// Replaces: every Conv2d(+folded BN)+SiLU, the C2f concat write, the
// Bottleneck residual add and Proto's ConvTranspose2d that upstream reaches through
// torch.nn.functional.conv2d / conv_transpose2d (call site: BscanBased/yolo8_seg_predict.py:8).
//
// GEMM view (roles chosen so that each lane's accumulators are CONSECUTIVE OUTPUT CHANNELS of one
// pixel, i.e. contiguous NHWC bytes):
//     D[channel][pixel] = sum_k  W[channel][k] * X[k][pixel],   k = (kh*KS + kw)*Cin + cin
//   A operand (MFMA rows)  = packed weights  [Cout_pad][Kpad]      (K contiguous)
//   B operand (MFMA cols)  = im2col gather of the NHWC input       (cin contiguous -> 16-byte chunks)
// v_mfma_f32_16x16x32_f16: lane l holds A[row l&15][k 8(l>>4)..+7], B[k 8(l>>4)..+7][col l&15],
// D[row 4(l>>4)+j][col l&15].
//
// Structure: PERSISTENT workgroups (CUs x resident blocks) walk the (pixel tile, channel tile) list;
// the loader is an independent cursor that runs two 64-deep K stages ahead of the MFMAs and crosses
// tile boundaries, so the LDS-DMA pipeline never drains and a tile's epilogue overlaps the next tile's
// first loads.  Both operands go HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4, one 16-byte chunk
// per lane, per-lane SOURCE address = the im2col gather; padded taps read a zero page); two LDS stages;
// one barrier per K step placed mid-step; fragment registers double-buffered so no MFMA waits on a
// just-issued ds_read.  LDS rows are 128 B (64 halves) with the 16-byte chunk index XOR-swizzled by
// (row & 7) -- applied on the source side (which chunk a lane fetches) and on the ds_read side, so the
// LDS-DMA image stays lane-linear; ds_read_b128 of a 16-row x 4-chunk fragment is conflict-free.
//
// Epilogue: + bias, SiLU, + residual, fp16 pack, 16-byte stores of 8 consecutive channels at a channel
// offset of a wider NHWC buffer (zero-copy concat); weight rows are permuted by the loader so that
// lane group g owns channels {g*8..g*8+7} (+32): per store instruction every pixel receives one
// contiguous 64-byte run.
#include <stdlib.h>

#include "common.h"

namespace m355 {

namespace {

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

// v * sigmoid(v);  exp2-based, rcp approx (1 ulp) -- far inside fp16 output rounding (shared with the fast epilogue)
__device__ __forceinline__ float silu_f(float v) { return m355_silu(v); }

// q = n / d, r = n % d for 0 <= n < 2^24 via a float reciprocal estimate + exact integer correction.
__device__ __forceinline__ void fast_divmod(int n, int d, float inv_d, int& q, int& r) {
  q = (int)((float)n * inv_d);
  r = n - q * d;
  if (r < 0) { r += d; --q; }
  if (r >= d) { r -= d; ++q; }
}

// MT/NT: 16x16 MFMA tiles per wave along channels / pixels.  WCH/WPX: waves along channels / pixels.
// EPI selects the special epilogues that only one launch of the network uses, each in an instantiation of its own:
//   0 plain (every other launch), 1 head output conv + box decode (a.dec_preds), 2 phase conv + proto.cv3 (a.phase && a.w2).
// Compiled into the one 128x128 1x1 kernel, the decode epilogue's register peak made the allocator spill 5 VGPRs at the top of
// EVERY 1x1 launch (24 B of scratch per lane, stored whether or not the epilogue ran: 8 MB of scratch writes per launch beside
// 75 MB of tensors, PMC WRITE_SIZE) -- the dominant kernel family paid for an epilogue that is off by default.
template <int MT, int NT, int WCH, int WPX, int KS, int EPI = 0>
__global__ __launch_bounds__(WCH * WPX * 64, (WCH * WPX == 8 ? 4 : 2)) void conv_igemm_kernel(const ConvArgs a) {  // (threads, min waves per SIMD)
  constexpr int NW = WCH * WPX;       // waves per workgroup: 4, or 8 (half-size wave tiles: four waves per SIMD with two
                                      // workgroups per CU hide the ~100-cycle issue cost of each LDS-DMA piece)
  static_assert(NW == 4 || NW == 8, "4 or 8 waves");
  constexpr int BCH = WCH * MT * 16;  // channel tile
  constexpr int BPX = WPX * NT * 16;  // pixel tile
  constexpr int ROWB = BK * 2;        // 128 bytes per LDS row
  constexpr int STAGE = (BCH + BPX) * ROWB;
  constexpr int A_IT = BPX / (NW * 8);  // LDS-DMA instructions per wave per stage, activations
  constexpr int W_IT = BCH / (NW * 8);  // weights
  static_assert(W_IT >= 1 && A_IT >= 1, "tile too small");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  unsigned long long st0 = 0, st1 = 0, st2 = 0, rt0 = 0;
  if (a.stamps) {
    st0 = __builtin_amdgcn_s_memtime();
    rt0 = __builtin_amdgcn_s_memrealtime();
  }
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // ---- tile list.  Blocks b, b+8, ... share an XCD (and its L2): logical block id lb groups them so
  // that at any time the blocks of one XCD work on a contiguous run of tiles (channel tiles fastest):
  // the tiles that gather the same pixels and their halo neighbours hit the same L2.
  const int tiles_ch = (a.Cout + BCH - 1) / BCH;
  const int tiles_px = (a.M + BPX - 1) / BPX;
  const int total = tiles_ch * tiles_px;
  const int nwg = gridDim.x;
  int lb;
  {
    const int orig = blockIdx.x, xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    lb = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
  }
  // K steps of a tile.  phase == 3 (stride-2 dgrad, a channel tile inside one phase, one tile per block): a phase's weight rows hold
  // only the taps it has -- (1 + a)(1 + b) of the four window slots, compact -- and its K loop ends there: 9 tap slots over the four
  // phases instead of 16.
  int nk = a.Kpad / BK;
  if (KS == 2 && a.phase == 3) {
    const int q = ((lb % tiles_ch) * BCH) / a.convt_co;
    nk = (1 + (q >> 1)) * (1 + (q & 1)) * a.Cin / BK;
  }
  const int my_tiles = lb < total ? (total - lb + nwg - 1) / nwg : 0;
  if (my_tiles == 0) return;
  const int total_stages = my_tiles * nk;

  // ---- loader.  Each lane owns LDS slot (row = 8*i' + lane/8, slot = lane%8) of every 8-row group it
  // loads; the source chunk for that slot is slot ^ (row & 7) = (lane&7) ^ (lane>>3).
  const int lrow = lane >> 3;
  const int cc = (lane & 7) ^ lrow;  // this lane's K-chunk column (8 halves) within a K step
  const int HoWo = a.Ho * a.Wo;
  const float inv_howo = 1.0f / (float)HoWo, inv_wo = 1.0f / (float)a.Wo;
  const float inv_cin = 1.0f / (float)a.Cin;
  const float inv_tch = 1.0f / (float)tiles_ch;

  const bool lin1 = KS == 1 && a.stride == 1 && a.pad == 0 && a.x_bstride == (long)HoWo * a.ldx;
  // Offsets in units of 8 elements (16 bytes; every stride is a multiple of 8, checked by the launcher): 32-bit registers.  As
  // 64-bit element offsets the two arrays cost 8 more VGPRs in the 128x128 1x1 kernel, which then spilled 5 registers to
  // scratch -- 8 MB of scratch writes per launch on top of 75 MB of tensors (PMC WRITE_SIZE), 35 registers / 140 MB in the
  // composed Proto kernel.
  int rowoff[A_IT];         // (b, hi0, wi0, 0) for each of this lane's pixel rows (may be negative: padding)
  int rowoff2[KS == 1 ? A_IT : 1];    // upsample read-through: (b, h >> 1, w >> 1, 0) in x2
  unsigned rowmask[A_IT];   // bit t: tap t is inside the image (and the row is < M)
  const half_t* wsrc[W_IT]; // this lane's weight rows (+ chunk column)
  int ld_tile = lb;         // loader cursor: tile, K step within the tile, global stage count
  int ld_t = 0, ld_g = 0;
  int ld_qb = 0;            // phase == 3: column parity b of the loader tile's phase (taps per window row = 1 + b)

  auto loader_setup = [&](int tile) __attribute__((always_inline)) {
    int tile_px, tile_ch;
    fast_divmod(tile, tiles_ch, inv_tch, tile_px, tile_ch);
    const int px_base = tile_px * BPX, ch_base = tile_ch * BCH;
#pragma clang loop unroll(full)
    for (int i = 0; i < A_IT; ++i) {
      const int prow = wave * (BPX / NW) + i * 8 + lrow;
      const int m = px_base + prow;
      const bool mv = m < a.M;
      const int mm = mv ? m : 0;
      if (KS == 1 && a.csplit > 0) {
        int b2, pix2, ho2, wo2;
        fast_divmod(mm, HoWo, inv_howo, b2, pix2);
        fast_divmod(pix2, a.Wo, inv_wo, ho2, wo2);
        rowoff2[KS == 1 ? i : 0] = (int)(((long)b2 * a.x2_bstride + ((long)(ho2 >> 1) * (a.Wo >> 1) + (wo2 >> 1)) * a.ldx2) >> 3);
      }
      if (KS == 1 && lin1) {   // 1x1 / stride 1 over contiguous images: the pixel index is the row index
        rowoff[i] = (int)(((long)mm * a.ldx) >> 3);
        rowmask[i] = mv ? 1u : 0u;
        continue;
      }
      int b, pix, ho, wo;
      fast_divmod(mm, HoWo, inv_howo, b, pix);
      fast_divmod(pix, a.Wo, inv_wo, ho, wo);
      unsigned mk = 0;
      int pad_y = a.pad, pad_x = a.pad;
      if (KS == 2 && a.phase) {   // phase conv: the 2x2 window of phase (py, px) starts at (h - 1 + py, w - 1 + px)
        const int q = ch_base / a.convt_co;
        pad_y = a.phase >= 2 ? 0 : 1 - (q >> 1);   // (phase >= 2, the stride-2 dgrad: every phase's window starts at (h, w))
        pad_x = a.phase >= 2 ? 0 : 1 - (q & 1);
        ld_qb = q & 1;
      }
      const int hi0 = ho * a.stride - pad_y;
      const int wi0 = wo * a.stride - pad_x;
      long off = (long)b * a.x_bstride + ((long)hi0 * a.Wi + wi0) * a.ldx;
      // NOTE: keep this loop body free of inner loops / continue: hipcc then fails to unroll the row loop for
      // A_IT = 8 and sends rowoff[] / rowmask[] to scratch (5x slower kernel).
      if (KS == 1) {
        mk = 1u;
      } else {
        // bit kh*KS+kw: tap inside the image
        const unsigned r0 = (unsigned)hi0 < (unsigned)a.Hi ? 1u : 0u, r1 = (unsigned)(hi0 + 1) < (unsigned)a.Hi ? 2u : 0u;
        const unsigned c0 = (unsigned)wi0 < (unsigned)a.Wi ? 1u : 0u, c1 = (unsigned)(wi0 + 1) < (unsigned)a.Wi ? 2u : 0u;
        if (KS == 3) {
          unsigned rv = r0 | r1 | ((unsigned)(hi0 + 2) < (unsigned)a.Hi ? 4u : 0u);
          unsigned cv = c0 | c1 | ((unsigned)(wi0 + 2) < (unsigned)a.Wi ? 4u : 0u);
          // tmode = dgrad of a stride-2 conv: output pixel (ho, wo) is the forward INPUT pixel; tap (kh, kw)
          // reads dY[(ho + 1 - kh) / 2][(wo + 1 - kw) / 2] when the parities match.  For valid taps
          // (ho + 1 - kh) / 2 == ((ho + 1) >> 1) - (kh >> 1), so the address stays base + per-tap offset.
          const int hb = (ho + 1) >> 1, wb = (wo + 1) >> 1;
          const unsigned h_in = (unsigned)hb < (unsigned)a.Hi ? 1u : 0u, h_in1 = (unsigned)(hb - 1) < (unsigned)a.Hi ? 1u : 0u;
          const unsigned w_in = (unsigned)wb < (unsigned)a.Wi ? 1u : 0u, w_in1 = (unsigned)(wb - 1) < (unsigned)a.Wi ? 1u : 0u;
          const unsigned rvt = ((ho + 1) & 1) ? (h_in << 1) : (h_in | (h_in1 << 2));
          const unsigned cvt = ((wo + 1) & 1) ? (w_in << 1) : (w_in | (w_in1 << 2));
          const bool tm = a.tmode != 0;
          rv = tm ? rvt : rv;
          cv = tm ? cvt : cv;
          off = tm ? ((long)b * a.x_bstride + ((long)hb * a.Wi + wb) * a.ldx) : off;
          mk = ((rv & 1u) ? cv : 0u) | ((rv & 2u) ? (cv << 3) : 0u) | ((rv & 4u) ? (cv << 6) : 0u);
        } else {
          const unsigned cv = c0 | c1;
          mk = (r0 ? cv : 0u) | (r1 ? (cv << 2) : 0u);
        }
      }
      if (!mv) mk = 0;
      rowoff[i] = (int)(off >> 3);          // arithmetic shift: off is a multiple of 8, negative for the top / left padding
      rowmask[i] = mk;
    }
    // weights: LDS row R (tile-local) holds the weight row of the channel the MFMA row maps to, so that
    // lane group g ends up owning 8 consecutive channels (see epilogue).  The permutation is applied on
    // the SOURCE side; LDS rows stay in plain MFMA-tile order (conflict-free ds_read_b128).
#pragma unroll
    for (int i = 0; i < W_IT; ++i) {
      const int R = wave * (BCH / NW) + i * 8 + lrow;
      const int blk = R / (MT * 16), Rl = R % (MT * 16);
      const int mt = Rl >> 4, r = Rl & 15;
      int chl;
      if (MT >= 2)
        chl = (mt >> 1) * 32 + (r >> 2) * 8 + (mt & 1) * 4 + (r & 3);
      else
        chl = r;
      wsrc[i] = a.w + (long)(ch_base + blk * MT * 16 + chl) * a.Kpad + cc * 8;
    }
  };

  // issue the LDS-DMA of the stage under the loader cursor into buffer (ld_g & 1), advance the cursor
  auto stage_next = [&]() __attribute__((always_inline)) {
    const int t = ld_t;
    char* sb = smem + (ld_g & 1) * STAGE;
    const bool skip_w = (a.dbg & 2) && ld_g > 1, skip_a = (a.dbg & 1) && ld_g > 1;
#pragma unroll
    for (int i = 0; i < W_IT; ++i) {
      if (!skip_w) glds16(wsrc[i] + t * BK, sb + (wave * (BCH / NW) + i * 8) * ROWB);
    }
    // activations: tap / cin of this lane's chunk at K step t
    const int kq = t * BK + cc * 8;
    int tap, cin, tapoff;
    if (KS == 1) {
      tap = kq >= a.Cin ? 31 : 0;
      tapoff = kq;
    } else {
      tap = (int)(((float)kq + 0.5f) * inv_cin);
      cin = kq - tap * a.Cin;
      int kh = (KS == 3) ? ((tap * 11) >> 5) : (tap >> 1);  // tap / KS for the few taps that exist
      int kw = tap - kh * KS;
      if (KS == 2 && a.phase == 3) {   // compact taps of the phase: 1 + b per window row
        kh = ld_qb ? (tap >> 1) : tap;
        kw = ld_qb ? (tap & 1) : 0;
        tap = 2 * kh + kw;             // the bit of the window slot in rowmask
      }
      tapoff = a.tmode ? (cin - ((kh >> 1) * a.Wi + (kw >> 1)) * a.ldx) : ((kh * a.Wi + kw) * a.ldx + cin);
    }
    char* ab = sb + BCH * ROWB;
#pragma clang loop unroll(full)
    for (int i = 0; i < A_IT; ++i) {
      const bool ok = (rowmask[i] >> tap) & 1u;
      const half_t* src = ok ? (a.x + ((long)rowoff[i] << 3) + tapoff) : a.zero;
      if (KS == 1 && a.csplit > 0 && ok && kq < a.csplit) src = a.x2 + ((long)rowoff2[KS == 1 ? i : 0] << 3) + kq;
      if (!skip_a) glds16(src, ab + (wave * (BPX / NW) + i * 8) * ROWB);
    }
    ++ld_g;
    if (++ld_t == nk) {
      ld_t = 0;
      ld_tile += nwg;
      if (ld_tile < total) loader_setup(ld_tile);
    }
  };

  // ---- fragment read addresses (byte offsets inside a stage)
  const int wch = wave / WPX, wpx = wave % WPX;
  const int l15 = lane & 15, g = lane >> 4;
  int aoff[MT];  // weights (A operand): byte offset of (row, chunk g) at ks = 0; ks = 1 is ^ 64
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int r = wch * MT * 16 + mt * 16 + l15;
    aoff[mt] = r * ROWB + ((g ^ (r & 7)) << 4);
  }
  int boff[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int r = wpx * NT * 16 + nt * 16 + l15;
    boff[nt] = BCH * ROWB + r * ROWB + ((g ^ (r & 7)) << 4);
  }

  float4v acc[MT][NT];
  half8 af0[MT], bf0[NT], af1[MT], bf1[NT];

  // Biases in LDS (a.bias_lds floats behind the two stages; launch_variant sized the allocation): a persistent block reads
  // the bias of every tile it walks, and an ordinary VGPR load issued while LDS-DMA is in flight makes hipcc wait vmcnt(0)
  // at its first use -- that drained the next tile's two prefetched stages at the top of every epilogue.
  float* const bias_s = (float*)(smem + 2 * STAGE);
  for (int i = tid; i < a.bias_lds; i += NW * 64) bias_s[i] = a.bias[i];
  // ---- pipeline prologue: two stages in flight, fragments ks=0 of stage 0 in registers
  loader_setup(ld_tile);
  stage_next();
  if (total_stages > 1) {
    stage_next();
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(W_IT + A_IT) : "memory");  // stage 0 landed, stage 1 in flight
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) af0[mt] = *(const half8*)(smem + aoff[mt]);
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) bf0[nt] = *(const half8*)(smem + boff[nt]);

  // One 64-deep K step of global stage gs (buffer gs&1), software pipelined, barrier MID-step:
  //   top : issue ds_reads ks=1(gs)   ->  16 MFMA ks=0(gs)          (read latency hidden)
  //   mid : vmcnt(0)+lgkmcnt(0)+barrier => stage gs+1 has landed for every wave and every wave is
  //         done reading stage gs -> issue ds_reads ks=0(gs+1), LDS-DMA for stage gs+2 into buffer
  //         gs&1 -> 16 MFMA ks=1(gs)
  // after_stores: this is the first K step of a tile that follows a fast epilogue in the same block (persistent walk).  That
  // epilogue's NST stores per lane were issued AFTER the two stages this step needs (vmcnt retires in issue order), so the
  // wait leaves them in flight instead of draining them: ~1-2 us per tile on the 1x1 layers.
  constexpr int NST = (MT >= 2 ? MT / 2 : 1) * NT;
  int gs = 0;
  auto kstep = [&](const bool after_stores) __attribute__((always_inline)) {
    const char* sb = smem + (gs & 1) * STAGE;
    const char* sn = smem + ((gs + 1) & 1) * STAGE;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) af1[mt] = *(const half8*)(sb + (aoff[mt] ^ 64));
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bf1[nt] = *(const half8*)(sb + (boff[nt] ^ 64));
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af0[mt], bf0[nt], acc[mt][nt], 0, 0, 0);
    if (after_stores) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NST) : "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if (!(a.dbg & 8)) __builtin_amdgcn_s_barrier();
    // ks=0 fragments of the next stage (after the very last stage this reads stale bytes, never used)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) af0[mt] = *(const half8*)(sn + aoff[mt]);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bf0[nt] = *(const half8*)(sn + boff[nt]);
    if (ld_g < total_stages) stage_next();
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af1[mt], bf1[nt], acc[mt][nt], 0, 0, 0);
    ++gs;
  };

  constexpr int GROUPS = (MT >= 2) ? MT / 2 : 1;  // 8-channel (MT>=2) or 4-channel (MT==1) groups per lane
  constexpr int GW = (MT >= 2) ? 8 : 4;

  bool stores_pending = false;
  for (int tile = lb; tile < total; tile += nwg) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = float4v{0.f, 0.f, 0.f, 0.f};
    int tile_px, tile_ch;
    fast_divmod(tile, tiles_ch, inv_tch, tile_px, tile_ch);
    const int px_base = tile_px * BPX, ch_base = tile_ch * BCH;
    // Fast epilogue (common.h): the whole tile is inside the tensor and images are contiguous, so the output address
    // is affine in the pixel index; the bias is fetched before the K loop and waits in registers.
    const bool fast = MT >= 2 && !a.out_f32 && a.convt_co == 0 && px_base + BPX <= a.M && ch_base + BCH <= a.Cout &&
                      a.y_bstride == (long)HoWo * a.ldy && (!a.res || a.r_bstride == (long)HoWo * a.ldr) && !(a.dbg & (32 | 256));
    // ConvTranspose 2x2 / s2 (virtual channel q * Co + co, q = dy * 2 + dx): when a channel tile lies inside one q
    // (Co a multiple of the tile) and the 16-pixel groups do not straddle image rows, (dy, dx) are uniform per block
    // and each group's 16 pixels go to 16 output pixels two apart in one row
    const bool fast_t = MT >= 2 && !a.out_f32 && a.convt_co > 0 && a.convt_co % BCH == 0 && a.Wo % 16 == 0 && !a.res &&
                        px_base + BPX <= a.M && ch_base + BCH <= a.Cout && !(a.dbg & (32 | 256));
    float4v bv[MT >= 2 ? MT / 2 : 1][2];
    if ((fast || fast_t) && !a.bias_lds) {
      const int cb0 = fast_t ? ch_base % a.convt_co : ch_base;   // the bias is indexed by the real output channel
      const float* bp = a.bias + cb0 + wch * MT * 16 + g * 8;
#pragma unroll
      for (int sg = 0; sg < MT / 2; ++sg) {
        bv[sg][0] = *(const float4v*)(bp + sg * 32);
        bv[sg][1] = *(const float4v*)(bp + sg * 32 + 4);
      }
    }
    if (a.stamps && tile == lb) st1 = __builtin_amdgcn_s_memtime();
    kstep(stores_pending);
    for (int t = 1; t < nk; ++t) kstep(false);
    stores_pending = false;
    if ((fast || fast_t) && a.bias_lds) {   // (every wave passed the prologue barrier after the staging loop)
      const int cb0 = fast_t ? ch_base % a.convt_co : ch_base;
      const float* bp = bias_s + cb0 + wch * MT * 16 + g * 8;
#pragma unroll
      for (int sg = 0; sg < MT / 2; ++sg) {
        bv[sg][0] = *(const float4v*)(bp + sg * 32);
        bv[sg][1] = *(const float4v*)(bp + sg * 32 + 4);
      }
    }
    if (a.stamps && tile == lb) st2 = __builtin_amdgcn_s_memtime();

    // ---- epilogue of this tile (the next tile's first two stages are already in flight / landed)
    const float* bias_fin = a.bias;   // bias of the conv whose output is stored
    if (KS != 2 && MT == 4 && !a.phase && a.w2) {
      // Conv + the 1x1 conv that is its only consumer (C2f.cv1 after a stride-2 conv) in one launch: the tile holds ALL
      // channels of its pixels (Cout == BCH == cout2, checked by the launcher), so Z = SiLU(conv + bias) goes to LDS as fp16
      // [pixel][BCH] instead of HBM, the BCH x BCH weights of the 1x1 conv beside it, and every wave multiplies its own
      // 64 ch x (NT x 16) px block with K = BCH in the K order of the stand-alone 1x1 launch (same bits).  acc is replaced
      // by the result; the ordinary epilogue below stores it with the second bias.
      constexpr int ROW2 = BCH * 2, CH2 = BCH / 8;          // bytes and 16-byte chunks per LDS row
      __syncthreads();                                      // every wave is done with the K-loop stages
      char* const zb = smem;                                // BPX rows
      char* const wb2 = smem + BPX * ROW2;                  // BCH rows: LDS row R <- logical channel chl(R), as the loader does
      for (int i = tid; i < BCH * CH2; i += NW * 64) {
        const int R = i / CH2, c = i - R * CH2;
        const int blk = R >> 6, Rl = R & 63, mt2 = Rl >> 4, r = Rl & 15;
        const int chl = blk * 64 + (mt2 >> 1) * 32 + (r >> 2) * 8 + (mt2 & 1) * 4 + (r & 3);
        *(float4v*)(wb2 + R * ROW2 + ((c ^ (R & (CH2 - 1))) << 4)) = *(const float4v*)(a.w2 + (long)chl * BCH + c * 8);
      }
      const float* bp = a.bias + wch * 64 + g * 8;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int p = wpx * NT * 16 + nt * 16 + l15;        // pixel inside the tile
#pragma unroll
        for (int sg = 0; sg < 2; ++sg) {
          const float4v b0 = *(const float4v*)(bp + sg * 32), b1 = *(const float4v*)(bp + sg * 32 + 4);
          half8 o;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float v0 = acc[2 * sg][nt][j] + b0[j], v1 = acc[2 * sg + 1][nt][j] + b1[j];
            if (a.act) { v0 = silu_f(v0); v1 = silu_f(v1); }
            o[j] = m355_to_half(v0);
            o[4 + j] = m355_to_half(v1);
          }
          const int c = (wch * 64 + sg * 32 + g * 8) >> 3;
          *(half8*)(zb + p * ROW2 + ((c ^ (p & (CH2 - 1))) << 4)) = o;
        }
      }
      __syncthreads();
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = float4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < BCH / 32; ++ks) {
        half8 a2[MT], b2[NT];
        const int c = ks * 4 + g;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const int R = wch * 64 + mt * 16 + l15;
          a2[mt] = *(const half8*)(wb2 + R * ROW2 + ((c ^ (R & (CH2 - 1))) << 4));
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const int p = wpx * NT * 16 + nt * 16 + l15;
          b2[nt] = *(const half8*)(zb + p * ROW2 + ((c ^ (p & (CH2 - 1))) << 4));
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2[mt], b2[nt], acc[mt][nt], 0, 0, 0);
      }
      __syncthreads();                                      // the stages are reused by the next tile (persistent mode)
      bias_fin = a.bias2;
      if (fast) {
        const float* bq = a.bias2 + wch * MT * 16 + g * 8;
#pragma unroll
        for (int sg = 0; sg < MT / 2; ++sg) {
          bv[sg][0] = *(const float4v*)(bq + sg * 32);
          bv[sg][1] = *(const float4v*)(bq + sg * 32 + 4);
        }
      }
    }
    if (EPI == 1 && KS == 1 && MT == 4 && NT == 4 && WCH == 2 && a.dec_preds) {
      // Head output conv + decode (the launcher checked: fp32 out, one channel tile, (wi + wo) * 64 floats fit the stages).
      // Two passes of 64 pixels = the 64-anchor blocks of head_decode_kernel, same arithmetic in the same order: the raw
      // row of a pixel goes to LDS instead of (or besides) HBM, four threads per anchor decode it, the prediction rows
      // leave through LDS with coalesced stores.
      const int nc = a.dec_nc, nm = a.dec_nm, wi = 64 + nc + nm, wo = 4 + nc + nm;
      float* const sin = (float*)smem;              // [64][wi]
      float* const sout = sin + 64 * wi;            // [64][wo]
#pragma unroll 1
      for (int half = 0; half < 2; ++half) {
        __syncthreads();                            // the stages / the previous pass are no longer read
        if (wpx == half) {
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            float* rp = sin + (nt * 16 + l15) * wi;
#pragma unroll
            for (int sg = 0; sg < 2; ++sg) {
              const int ch0 = wch * 64 + sg * 32 + g * 8;
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                if (ch0 + j < wi) rp[ch0 + j] = acc[2 * sg][nt][j] + a.bias[ch0 + j];
                if (ch0 + 4 + j < wi) rp[ch0 + 4 + j] = acc[2 * sg + 1][nt][j] + a.bias[ch0 + 4 + j];
              }
            }
          }
        }
        __syncthreads();
        const int al_blk = tid >> 2, q = tid & 3;
        const int m = px_base + half * 64 + al_blk;
        const bool valid = m < a.M;
        const float* rp = sin + al_blk * wi;
        float v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = rp[q * 16 + j];
        float mx = v[0];
#pragma unroll
        for (int j = 1; j < 16; ++j) mx = fmaxf(mx, v[j]);
        float se = 0.f, sw = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const float e = __expf(v[j] - mx);
          se += e;
          sw += e * (float)j;
        }
        const float d = sw / se;
        int bb, al;
        fast_divmod(valid ? m : 0, HoWo, inv_howo, bb, al);
        int gy, gx;
        fast_divmod(al, a.Wo, inv_wo, gy, gx);
        const float ax = (float)gx + 0.5f, ay = (float)gy + 0.5f;
        const int lbase = lane & ~3;
        const float dl = __shfl(d, lbase + 0), dt = __shfl(d, lbase + 1), dr = __shfl(d, lbase + 2), db = __shfl(d, lbase + 3);
        const float x1 = ax - dl, y1 = ay - dt, x2 = ax + dr, y2 = ay + db;
        float o;
        if (q == 0) o = (x1 + x2) * 0.5f;
        else if (q == 1) o = (y1 + y2) * 0.5f;
        else if (q == 2) o = x2 - x1;
        else o = y2 - y1;
        float* pp = sout + al_blk * wo;
        pp[q] = o * a.dec_stride;
        for (int j = q; j < nc; j += 4) {
          const float z = rp[64 + j];
          pp[4 + j] = 1.0f / (1.0f + __expf(-z));
        }
        for (int j = q; j < nm; j += 4) pp[4 + nc + j] = rp[64 + nc + j];
        __syncthreads();
        const float inv_wo2 = 1.0f / (float)wo;
        for (int i = tid; i < 64 * wo; i += NW * 64) {
          int r, j;
          fast_divmod(i, wo, inv_wo2, r, j);
          const int mr = px_base + half * 64 + r;
          if (mr >= a.M) continue;
          int b2, al2;
          fast_divmod(mr, HoWo, inv_howo, b2, al2);
          a.dec_preds[((long)b2 * a.dec_A + a.dec_level_off + al2) * wo + j] = sout[i];
        }
      }
      __syncthreads();                              // (persistent mode: the stages are reused by the next tile)
      if (!a.dec_keep_raw) continue;
    }
    if (EPI == 2 && KS == 2 && MT == 4 && NT == 4 && WCH == 2) {   // (the launcher checked a.phase && a.w2)
      // Phase conv + proto.cv3 in one epilogue.  The 128 ch x 128 px tile Z = SiLU(phase conv) goes to LDS as fp16
      // [pixel][128 ch] (16-byte chunk c of pixel p in slot c ^ (p & 15): conflict-free for the writes below and for the
      // B-fragment reads), the 32 x 128 weights of the 1x1 conv likewise [row][128] with the usual row permutation;
      // every wave then multiplies 32 channels x 32 pixels x K = 128 (16 MFMAs) and stores 8 channels per lane.
      const int q = ch_base / a.convt_co, dy = q >> 1, dx = q & 1;
      __syncthreads();                                   // every wave is done with the K-loop stages
      char* const zb = smem;                             // 128 px x 256 B
      char* const wb2 = smem + 128 * 256;                // 32 rows x 256 B
      for (int i = tid; i < 32 * 16; i += 256) {         // weights: LDS row R <- logical channel chl(R), chunk swizzled by R
        const int R = i >> 4, c = i & 15;
        const int mt2 = R >> 4, r = R & 15;
        const int chl = (r >> 2) * 8 + mt2 * 4 + (r & 3);
        *(float4v*)(wb2 + R * 256 + ((c ^ (R & 15)) << 4)) = *(const float4v*)(a.w2 + (long)chl * a.convt_co + c * 8);
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int p = wpx * NT * 16 + nt * 16 + l15;     // pixel inside the tile
        const int m = px_base + p;
        const int mm = m < a.M ? m : a.M - 1;
        int bb, pix, ho, wo;
        fast_divmod(mm, HoWo, inv_howo, bb, pix);
        fast_divmod(pix, a.Wo, inv_wo, ho, wo);
        const int Y = 2 * ho + dy, X = 2 * wo + dx;
        const int ry = Y == 0 ? 0 : (Y == 2 * a.Ho - 1 ? 2 : 1), rx = X == 0 ? 0 : (X == 2 * a.Wo - 1 ? 2 : 1);
        const float* bp = a.bias + (ry * 3 + rx) * a.convt_co + wch * 64 + g * 8;
#pragma unroll
        for (int sg = 0; sg < 2; ++sg) {
          const float4v b0 = *(const float4v*)(bp + sg * 32), b1 = *(const float4v*)(bp + sg * 32 + 4);
          half8 o;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float v0 = acc[2 * sg][nt][j] + b0[j], v1 = acc[2 * sg + 1][nt][j] + b1[j];
            if (a.act) { v0 = silu_f(v0); v1 = silu_f(v1); }
            o[j] = m355_to_half(v0);
            o[4 + j] = m355_to_half(v1);
          }
          const int c = (wch * 64 + sg * 32 + g * 8) >> 3;   // 16-byte chunk index of these 8 channels
          *(half8*)(zb + p * 256 + ((c ^ (p & 15)) << 4)) = o;
        }
      }
      __syncthreads();
      float4v acc2[2][2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc2[i][j] = float4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {                   // K = 128 channels in four 32-deep slices
        half8 a2[2], b2[2];
        const int c = ks * 4 + g;                        // chunk of this lane's 8 k values
#pragma unroll
        for (int mt2 = 0; mt2 < 2; ++mt2) {
          const int R = mt2 * 16 + l15;
          a2[mt2] = *(const half8*)(wb2 + R * 256 + ((c ^ (R & 15)) << 4));
        }
#pragma unroll
        for (int nt2 = 0; nt2 < 2; ++nt2) {
          const int p = wave * 32 + nt2 * 16 + l15;
          b2[nt2] = *(const half8*)(zb + p * 256 + ((c ^ (p & 15)) << 4));
        }
#pragma unroll
        for (int mt2 = 0; mt2 < 2; ++mt2)
#pragma unroll
          for (int nt2 = 0; nt2 < 2; ++nt2)
            acc2[mt2][nt2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2[mt2], b2[nt2], acc2[mt2][nt2], 0, 0, 0);
      }
      // Store.  Lane (g, l15) holds channels 8g..8g+7 of pixel l15: stored from there, the four 16-byte pieces of a pixel's
      // 64-byte row leave from lanes 16 apart and reach memory as four separate partial writes (PMC: 189 MB written per launch
      // for 52 MB of prototypes).  The rows go through the wave's own Z rows in LDS (its MFMAs above were their only readers)
      // and come back as lane = 4 * pixel + piece: four adjacent lanes = one whole row per request.
      const float4v c0 = *(const float4v*)(a.bias2 + g * 8), c1 = *(const float4v*)(a.bias2 + g * 8 + 4);
      char* const ob = zb + wave * 32 * 256;             // 32 px x 64 B, piece c of pixel pl in slot c ^ ((pl >> 2) & 3)
#pragma unroll
      for (int nt2 = 0; nt2 < 2; ++nt2) {
        const int pl = nt2 * 16 + l15;
        half8 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float v0 = acc2[0][nt2][j] + c0[j], v1 = acc2[1][nt2][j] + c1[j];
          v0 = silu_f(v0);
          v1 = silu_f(v1);
          o[j] = m355_to_half(v0);
          o[4 + j] = m355_to_half(v1);
        }
        *(half8*)(ob + pl * 64 + ((g ^ ((pl >> 2) & 3)) << 4)) = o;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the wave's own LDS writes have landed (same-wave exchange: no barrier)
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int qd = it * 64 + lane, pl = qd >> 2, c = qd & 3;
        const half8 o = *(const half8*)(ob + pl * 64 + ((c ^ ((pl >> 2) & 3)) << 4));
        const int m = px_base + wave * 32 + pl;
        if (m >= a.M) continue;
        int bb, pix, ho, wo;
        fast_divmod(m, HoWo, inv_howo, bb, pix);
        fast_divmod(pix, a.Wo, inv_wo, ho, wo);
        const int Y = 2 * ho + dy, X = 2 * wo + dx;
        *(half8*)((half_t*)a.y + (long)bb * a.y_bstride + ((long)Y * (2 * a.Wo) + X) * a.ldy + c * 8) = o;
      }
      __syncthreads();                                   // (persistent mode: the stages are reused by the next tile)
      continue;
    }
    // phase conv: pixel-shuffle store, bias by border class of the output pixel.  The stride-2 dgrad form (phase == 2) comes here
    // only to accumulate into a gradient slice that already holds a consumer's contribution (a.res); without one it takes the
    // ConvTranspose fast stores below, and with several phases per channel tile (convt_co < tile) the generic epilogue.
    if (KS == 2 && MT >= 2 && a.phase && !(a.phase >= 2 && (fast_t || a.convt_co % BCH))) {
      const int q = ch_base / a.convt_co, dy = q >> 1, dx = q & 1;
      const int co = ch_base - q * a.convt_co + wch * MT * 16 + g * 8;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int m = px_base + wpx * NT * 16 + nt * 16 + l15;
        if (m >= a.M) continue;
        int bb, pix, ho, wo;
        fast_divmod(m, HoWo, inv_howo, bb, pix);
        fast_divmod(pix, a.Wo, inv_wo, ho, wo);
        const int Y = 2 * ho + dy, X = 2 * wo + dx;
        const int ry = Y == 0 ? 0 : (Y == 2 * a.Ho - 1 ? 2 : 1), rx = X == 0 ? 0 : (X == 2 * a.Wo - 1 ? 2 : 1);
        const float* bp = a.bias + (a.phase >= 2 ? 0 : (ry * 3 + rx) * a.convt_co) + co;
        half_t* yp = (half_t*)a.y + (long)bb * a.y_bstride + ((long)Y * (2 * a.Wo) + X) * a.ldy + co;
        const half_t* rp = a.res ? a.res + (long)bb * a.r_bstride + ((long)Y * (2 * a.Wo) + X) * a.ldr + co : nullptr;
#pragma unroll
        for (int sg = 0; sg < MT / 2; ++sg) {
          if (co + sg * 32 >= a.convt_co) continue;
          const float4v b0 = *(const float4v*)(bp + sg * 32), b1 = *(const float4v*)(bp + sg * 32 + 4);
          half8 o, rv;
          if (rp) rv = *(const half8*)(rp + sg * 32);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float v0 = acc[2 * sg][nt][j] + b0[j], v1 = acc[(MT >= 2 ? 2 * sg + 1 : 0)][nt][j] + b1[j];
            if (a.act) { v0 = silu_f(v0); v1 = silu_f(v1); }
            if (rp) { v0 += (float)rv[j]; v1 += (float)rv[4 + j]; }
            o[j] = m355_to_half(v0);
            o[4 + j] = m355_to_half(v1);
          }
          *(half8*)(yp + sg * 32) = o;
        }
      }
      continue;
    }
    // stride-2 dgrad with 32 forward input channels (phase == 2, convt_co == 32): every 32-channel group of the tile is one phase of
    // the same 32 dX channels.  Whole tiles only; the generic epilogue below did this with two divisions per group at 0.74 TB/s.
    if (KS == 2 && MT >= 2 && a.phase == 2 && a.convt_co == 32 && !a.out_f32 && !a.res && !a.act && px_base + BPX <= a.M &&
        ch_base + BCH <= a.Cout && !(a.dbg & (32 | 256))) {
      const float4v b0 = *(const float4v*)(a.bias + g * 8), b1 = *(const float4v*)(a.bias + g * 8 + 4);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int m = px_base + wpx * NT * 16 + nt * 16 + l15;
        int bb, pix, ho, wo;
        fast_divmod(m, HoWo, inv_howo, bb, pix);
        fast_divmod(pix, a.Wo, inv_wo, ho, wo);
        half_t* const yb = (half_t*)a.y + (long)bb * a.y_bstride + ((long)(2 * ho) * (2 * a.Wo) + 2 * wo) * a.ldy + g * 8;
#pragma unroll
        for (int sg = 0; sg < MT / 2; ++sg) {
          const int q = (ch_base + wch * MT * 16 + sg * 32) >> 5;
          half8 o;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            o[j] = m355_to_half(acc[2 * sg][nt][j] + b0[j]);
            o[4 + j] = m355_to_half(acc[(MT >= 2 ? 2 * sg + 1 : 0)][nt][j] + b1[j]);
          }
          *(half8*)(yb + ((long)(q >> 1) * (2 * a.Wo) + (q & 1)) * a.ldy) = o;
        }
      }
      continue;
    }
    if (MT >= 2 && fast_t) {
      const int q = ch_base / a.convt_co, dy = q >> 1, dx = q & 1;
      const int co = ch_base - q * a.convt_co + wch * MT * 16 + g * 8;
      half_t* yp[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int m = px_base + wpx * NT * 16 + nt * 16 + l15;
        int bb, pix, ho, wo;
        fast_divmod(m, HoWo, inv_howo, bb, pix);
        fast_divmod(pix, a.Wo, inv_wo, ho, wo);
        yp[nt] = (half_t*)a.y + (long)bb * a.y_bstride + ((long)(2 * ho + dy) * (2 * a.Wo) + 2 * wo + dx) * a.ldy + co;
      }
      if (a.act) conv_epilogue_fast_ptrs<(MT >= 2 ? MT : 2), NT, true>(acc, bv, yp);
      else conv_epilogue_fast_ptrs<(MT >= 2 ? MT : 2), NT, false>(acc, bv, yp);
      continue;
    }
    if (MT >= 2 && fast) {
      const long m0 = px_base + wpx * NT * 16 + l15;
      const int cho = ch_base + wch * MT * 16 + g * 8;
      half_t* yp = (half_t*)a.y + m0 * a.ldy + cho;
      const long ystep = 16L * a.ldy;
      if (a.res) {
        const half_t* rp = a.res + m0 * a.ldr + cho;
        const long rstep = 16L * a.ldr;
        if (a.act) conv_epilogue_fast<(MT >= 2 ? MT : 2), NT, true, true>(acc, bv, yp, ystep, rp, rstep);
        else conv_epilogue_fast<(MT >= 2 ? MT : 2), NT, false, true>(acc, bv, yp, ystep, rp, rstep);
      } else {
        if (a.act) conv_epilogue_fast<(MT >= 2 ? MT : 2), NT, true, false>(acc, bv, yp, ystep, nullptr, 0);
        else conv_epilogue_fast<(MT >= 2 ? MT : 2), NT, false, false>(acc, bv, yp, ystep, nullptr, 0);
        stores_pending = a.bias_lds > 0 && !(a.dbg & 512);   // (no other vector-memory instruction between these stores and the next wait)
      }
      if (a.stamps && tid == 0 && tile == lb) {   // diagnostic builds only (tools/stamps_igemm.py)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned long long* o = a.stamps + (long)blockIdx.x * 8;
        o[0] = st0; o[1] = st1; o[2] = st2; o[3] = __builtin_amdgcn_s_memtime(); o[4] = rt0; o[5] = __builtin_amdgcn_s_memrealtime();
      }
      continue;
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int m = px_base + wpx * NT * 16 + nt * 16 + l15;
      if (m >= a.M) continue;
      int b, pix;
      fast_divmod(m, HoWo, inv_howo, b, pix);
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
        long yoff;
        int cidx = ch0;  // bias / output channel index
        if (a.convt_co > 0) {
          const int q = ch0 / a.convt_co;
          cidx = ch0 - q * a.convt_co;
          const int ho = pix / a.Wo, wo = pix - ho * a.Wo;
          yoff = (long)b * a.y_bstride + ((long)(2 * ho + (q >> 1)) * (2 * a.Wo) + 2 * wo + (q & 1)) * a.ldy + cidx;
        } else {
          yoff = (long)b * a.y_bstride + (long)pix * a.ldy + ch0;
        }
#pragma unroll
        for (int j = 0; j < GW; ++j) v[j] += bias_fin[cidx + j];
        if (a.act && !(a.dbg & 32)) {
#pragma unroll
          for (int j = 0; j < GW; ++j) v[j] = silu_f(v[j]);
        }
        if (a.res) {
          const half_t* rp = a.res + (long)b * a.r_bstride + (long)pix * a.ldr + ch0;
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
        if (a.out_f32) {
          float* yp = (float*)a.y + yoff;
          const int lim = (a.convt_co > 0) ? a.convt_co : a.Cout;
          if (cidx + GW <= lim) {   // whole group inside: 16-byte stores (rows of the raw head map are only 4-byte aligned)
            typedef float float4u __attribute__((ext_vector_type(4), aligned(4)));
            *(float4u*)yp = float4u{v[0], v[1], v[2], v[3]};
            if (GW == 8) *(float4u*)(yp + 4) = float4u{v[GW - 4], v[GW - 3], v[GW - 2], v[GW - 1]};
          } else {
#pragma unroll
            for (int j = 0; j < GW; ++j)
              if (cidx + j < lim) yp[j] = v[j];
          }
        } else {
          half_t* yp = (half_t*)a.y + yoff;
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
    }
  }
  // every LDS-DMA this wave issued was waited for by the last kstep's vmcnt(0): no DMA can land after exit
}

int g_num_cus = 0;

template <int MT, int NT, int WCH, int WPX>
int launch_variant(const ConvArgs& a, hipStream_t s) {
  constexpr int BCH = WCH * MT * 16, BPX = WPX * NT * 16;
  constexpr int LDS = 2 * (BCH + BPX) * BK * 2;
  const int tiles_ch = (a.Cout + BCH - 1) / BCH;
  const int tiles_px = (a.M + BPX - 1) / BPX;
  if (!conv_rows_covered(a, BCH)) return -1;   // the tile would fetch weight rows past the caller's buffer
  if (g_num_cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return -2;
    g_num_cus = prop.multiProcessorCount;
  }
  // One tile per block by default.  A persistent grid (CUs x resident blocks, loader cursor crossing
  // tile boundaries) is supported by the kernel but measured SLOWER on MI355X for every layer of this
  // network (fewer resident waves, epilogue-store drain on the next tile's first vmcnt(0)); it is kept
  // behind dbg bit 64 for experiments.
  int grid_x = tiles_ch * tiles_px;
  // measured again after the fast epilogue: neutral for the 128x128 tile, 7-20 % faster for 1x1 layers on the
  // narrow channel tiles (proto.cv3 63 -> 51 us), so those run persistent
  // round 2, with the biases in LDS and the counted wait after a tile's stores: in isolation the 128-channel 1x1 layers gain
  // 3-10 % from the persistent walk once they have more tiles than resident blocks (tools/conv1_sweep.py: 39.2 -> 34.2,
  // 47.4 -> 43.6, 21.1 -> 19.7 us), inside the network nothing (model.4.cv2 46.9 -> 47.8, model.15.cv2 37.9 -> 39.6 us):
  // still one tile per block for them
  if ((a.dbg & 64) || (a.ksize == 1 && BCH <= 64 && !knobs().no_persist)) {
    int per_cu = (160 * 1024) / LDS;
    if (per_cu > 2) per_cu = 2;
    if (grid_x > g_num_cus * per_cu) grid_x = g_num_cus * per_cu;
  }
  // bias table in LDS: the 1x1 / 3x3 fast-epilogue launches without the fused second conv or the decode (their epilogues
  // reuse the stage memory and read a.bias directly)
  ConvArgs aa = a;
  int lds_bytes = LDS;
  {
    const int nb = a.convt_co > 0 ? a.convt_co : tiles_ch * BCH;
    const bool plain = !a.w2 && !a.dec_preds && !a.phase && !a.out_f32 && MT >= 2 && !knobs().no_bias_lds;
    if (plain && nb * 4 <= 8192 && (a.w_rows == 0 || nb <= a.w_rows)) {
      aa.bias_lds = nb;
      lds_bytes += nb * 4;
    }
  }
  const dim3 grid(grid_x), block(WCH * WPX * 64);
  hipError_t e;
  auto k1 = conv_igemm_kernel<MT, NT, WCH, WPX, 1>;
  auto k2 = conv_igemm_kernel<MT, NT, WCH, WPX, 2>;
  auto k3 = conv_igemm_kernel<MT, NT, WCH, WPX, 3>;
  auto k = a.ksize == 1 ? k1 : (a.ksize == 2 ? k2 : k3);
  if constexpr (MT == 4 && NT == 4 && WCH == 2 && WPX == 2) {          // the special epilogues live in instantiations of their own
    if (a.dec_preds) {
      if (a.ksize != 1) return -1;
      k = conv_igemm_kernel<MT, NT, WCH, WPX, 1, 1>;
    } else if (a.phase && a.w2) {
      if (a.ksize != 2) return -1;
      k = conv_igemm_kernel<MT, NT, WCH, WPX, 2, 2>;
    }
  } else {
    if (a.dec_preds || (a.phase && a.w2)) return -1;
  }
  if (lds_bytes > 65536) {
    e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, LDS + 8192);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(k, grid, block, lds_bytes, s, aa);
  return (int)hipGetLastError();
}

}  // namespace

const Knobs& knobs() {
  static const Knobs k = [] {
    Knobs v{};
    v.no_fast_epi = getenv("M355_NO_FAST_EPI") != nullptr;
    v.no_wide = getenv("M355_NO_WIDE") != nullptr;
    v.no_m32 = getenv("M355_NO_M32") != nullptr;
    v.no_bias_lds = getenv("M355_NO_BIAS_LDS") != nullptr;
    v.static_tiles = getenv("M355_STATIC_TILES") != nullptr;   // persistent kernels: static tile walk instead of the queue
    v.no_persist = getenv("M355_NO_PERSIST") != nullptr;
    v.stem_gather = getenv("M355_STEM_GATHER") != nullptr;
    v.persist = getenv("M355_PERSIST") ? atoi(getenv("M355_PERSIST")) : 0;
    v.halo_variant = getenv("M355_HALO_VARIANT") ? atoi(getenv("M355_HALO_VARIANT")) : 2;
    v.smallm = getenv("M355_SMALLM") ? atoi(getenv("M355_SMALLM")) : 300;
    return v;
  }();
  return k;
}

int conv_cout_pad(int cout) { return (cout + 127) / 128 * 128; }
int conv_kpad(int cin, int ksize) { return (cin * ksize * ksize + BK - 1) / BK * BK; }

int conv_pick_tile(int cout, long M) {
  // small pixel counts (20x20 maps at batch 32): a 128x128 grid leaves most CUs with <= 1 block; halve the
  // channel tile to double the number of blocks
  const long thr = knobs().smallm;  // measured sweep 0/300/600/1000 on MI355X: 300 is best
  if (cout > 64 && M * ((cout + 127) / 128) / 128 < thr) return TILE_64x128;
  if (cout > 64) return TILE_128x128;
  if (cout > 32) return TILE_64x128;
  return TILE_32x256;
}

bool conv_forced_tile_extent(int tile, int cout, int* bch, int* bpx) {
  int c = 0, p = 0;
  switch (tile) {
    case TILE_128x128: c = 128; p = 128; break;
    case TILE_64x128: case TILE_64x128W8: c = 64; p = 128; break;
    case TILE_32x256: c = 32; p = 256; break;
    case TILE_64x256: c = 64; p = 256; break;
    case TILE_HALO: case TILE_HALO8W: case TILE_HALO4W: c = cout > 64 ? 128 : 64; p = 128; break;
    case TILE_HALOWIDE: c = 128; p = 256; break;
    case TILE_SLAB: c = 64; p = 256; break;
    case TILE_C32: c = 32; p = 256; break;
    case TILE_W1: c = 128; p = 64; break;
    case TILE_PLANES: c = 64; p = 256; break;
    case TILE_M32: c = cout > 64 ? 128 : 64; p = 256; break;
    case TILE_M32_128: c = 128; p = 128; break;
    case TILE_M32_64x16: c = 64; p = 256; break;
    case TILE_M32_64x8: c = 64; p = 128; break;
    default: return false;
  }
  if (bch) *bch = c;
  if (bpx) *bpx = p;
  return true;
}

int launch_conv_igemm(const ConvArgs& a0, int force_tile, hipStream_t s) {
  ConvArgs a = a0;
  if (knobs().no_fast_epi) a.dbg |= 256;
  if ((knobs().persist & (a.ksize == 1 ? 1 : 2)) && a.phase != 3) a.dbg |= 64;
  if (a.ksize < 1 || a.ksize > 3) return -1;
  if (a.ksize == 2 && !a.phase && (a.stride != 2 || a.pad != 0 || a.tmode)) return -1;  // the ConvT-dgrad form ...
  if (a.ksize == 2 && a.phase == 1 && (a.stride != 1 || a.tmode || a.out_f32 || a.convt_co <= 0 || a.convt_co % 64)) return -1;  // ... or a phase conv
  // ... or the dgrad of a 3x3 / stride-2 / pad-1 conv as four phase convs (phase == 2): x is dY, Cout = 4 * convt_co virtual channels
  // (phase q = 2 * (row parity) + column parity of the dX pixel, then the forward input channel), 2x2 windows starting at (h, w)
  if (a.ksize == 2 && a.phase >= 2 && (a.stride != 1 || a.pad != 0 || a.tmode || a.out_f32 || a.convt_co <= 0 || a.convt_co % 8 || a.w2 || a.act ||
                                        a.Cout != 4 * a.convt_co))
    return -1;
  // phase == 3: the same with COMPACT weight rows (a phase's K axis holds only its (1 + a)(1 + b) taps, its K loop ends there):
  // needs the channel tile inside one phase and one tile per block
  if (a.phase == 3 && (a.convt_co % 64 || a.Cin % BK || (a.dbg & 64))) return -1;
  if (a.phase < 0 || a.phase > 3 || (a.phase && a.ksize != 2)) return -1;
  if (a.Cin % 8 || a.ldx % 8 || (!a.out_f32 && (a.ldy % 8 || a.Cout % 8))) return -1;
  if (a.M >= (1 << 24)) return -1;  // fast_divmod range
  // the loader keeps row offsets as 32-bit counts of 8 elements: every stride a multiple of 8, the input within 2^34 elements
  if (a.x_bstride % 8 || (a.csplit > 0 && (a.x2_bstride % 8 || a.ldx2 % 8 || a.csplit % 8))) return -1;
  {
    const long nimg = a.Ho * a.Wo > 0 ? (a.M + (long)a.Ho * a.Wo - 1) / ((long)a.Ho * a.Wo) : 0;
    if (nimg * a.x_bstride >= (1L << 34) || (a.csplit > 0 && nimg * a.x2_bstride >= (1L << 34))) return -1;
  }
  int tile = force_tile & 0xff;
  if (force_tile < 0) tile = -1;
  if (tile < 0) tile = conv_pick_tile(a.Cout, a.M);
  if (a.ksize == 2 && a.phase == 1 && ((tile == TILE_128x128 && a.convt_co % 128) || (tile != TILE_128x128 && tile != TILE_64x128)))
    return -1;   // a channel tile must lie inside one phase
  if (a.ksize == 2 && a.phase >= 2) {   // a tile inside one phase, or whole phases inside a tile (then no accumulation: generic epilogue)
    if (force_tile < 0 && tile == TILE_128x128 && a.convt_co % 128 && a.convt_co % 64 == 0) tile = TILE_64x128;
    const int bch = tile == TILE_128x128 ? 128 : 64;
    if (tile != TILE_128x128 && tile != TILE_64x128) return -1;
    if (a.convt_co % bch && (bch % a.convt_co || a.res || a.phase == 3)) return -1;
  }
  if (a.w2 && a.phase && !(a.ksize == 2 && tile == TILE_128x128 && a.convt_co == 128 && a.cout2 == 32 && a.bias2)) return -1;
  if (a.dec_preds && !(a.ksize == 1 && a.out_f32 && tile == TILE_128x128 && a.Cout == 64 + a.dec_nc + a.dec_nm && a.Cout <= 128 &&
                       (a.Cout + 4 + a.dec_nc + a.dec_nm) * 64 * 4 <= 65536 && !a.res && a.convt_co == 0))
    return -1;
  // conv + following 1x1 in one launch: one channel tile holding every channel, square 1x1, fp16 out
  if (a.w2 && !a.phase &&
      !(a.ksize != 2 && a.bias2 && !a.out_f32 && a.convt_co == 0 && !a.tmode && a.cout2 == a.Cout &&
        ((tile == TILE_128x128 && a.Cout == 128) || (tile == TILE_64x128 && a.Cout == 64))))
    return -1;
  switch (tile) {
    case TILE_128x128: return launch_variant<4, 4, 2, 2>(a, s);
    case TILE_64x128: return launch_variant<4, 2, 1, 4>(a, s);
    case TILE_32x256: return launch_variant<2, 4, 1, 4>(a, s);
    case TILE_64x256: return launch_variant<4, 4, 1, 4>(a, s);
    // 8 waves x (32 ch x 32 px), four waves per SIMD: measured equal to the 4-wave tiles on every layer class
    // (tools/tile_sweep.py: the 1x1 layers run at 4.6 TB/s in isolation whatever the tile); kept for experiments
    case TILE_64x128W8: return launch_variant<2, 2, 2, 4>(a, s);
    // 8 waves x (64 ch x 64 px), 128x256 and 256x128 tiles (one block per CU, 0.19 instead of 0.25 KiB of LDS intake per
    // MFMA): 5-20 % slower than the 4-wave tiles on every 3x3 / stride-2 / 1x1 layer class of this network; removed
    default: return -1;
  }
}

}  // namespace m355
