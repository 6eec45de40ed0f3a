// 1x1 NHWC fp16 convolution (a plain GEMM over pixels) with the WEIGHTS IN REGISTERS (gfx950, v_mfma_f32_32x32x16_f16).
//
// Replaces: the 1x1 Conv+BN+SiLU layers (C2f.cv1 / C2f.cv2 / SPPF.cv1) that upstream reaches through
// torch.nn.functional.conv2d (call site: /root/reference/BscanBased/yolo8_seg_predict.py:8).
//
// Why.  The 1x1 family on the im2col kernel was the dominant family of the forward for two rounds at 0.28-0.31 of the HBM
// roof: every 128 x 128 tile streams its 128 weight rows through LDS-DMA again (as many bytes as its activations, at
// 60-180 issue cycles per KiB piece), reads every weight fragment back with a ds_read, and pays an 8 k-cycle prologue per
// tile.  For K <= 512 the 32 rows x K weights of one channel block are at most 128 VGPRs: here every wave keeps ITS
// channel block's fragments in registers for the whole life of a persistent block, only the activation tile goes through
// LDS (one DMA stream, one ds_read_b128 per MFMA), and the output leaves as 64-byte row segments through a per-wave
// transposition buffer.
//
// Block = 8 waves, one block per CU, CB = Cout / 32 channel blocks per tile (4 or 8).  Every wave = one channel block x
// PB 32-pixel MFMA blocks (PB = 2: 64 pixels): CB = 8 -> tile = 64 pixels, wave = channel block; CB = 4 -> tile = 128 pixels,
// wave = (channel block, pixel half); CB = 4, PB = 1 (K = 384: a 128-pixel tile of 768-byte rows does not fit two buffers) ->
// tile = 64 pixels.  Cout = 512 runs as two channel tiles of 256.  Pixels are the flattened (image, row, column) index: a 1x1
// convolution has no halo.
//
// SPLIT (round 4; model.15.cv1 of the s scale: Upsample + Concat + C2f.cv1): the first `csplit` channels of a pixel come from a
// tensor of half the resolution, pixel (h >> 1, w >> 1) -- nn.Upsample(scale_factor=2, 'nearest') read through instead of
// materialised -- and the rest from x at their own channel offset.  An LDS-DMA instruction takes ONE buffer descriptor, so the
// tile is kept as two images (256 read-through channels per pixel, then 128 own channels per pixel): every 1 KiB piece has one
// source.
//
// LDS image: one row of K fp16 per pixel; the low bits of the 16-byte chunk index are XOR-ed with the pixel index (4 bits
// when K % 128 == 0, else 3) on the DMA source side and on the reads, so the 32 lanes of a fragment read (32 pixels, same
// chunk) spread over the banks.
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace m355 {
namespace {

typedef float float16v __attribute__((ext_vector_type(16)));
constexpr int NWAVES = 8;

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, int voff, int soff, char* lds) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds, 16, voff, soff, 0, 0);
}
__device__ __forceinline__ int row_plain(int rho) { return 16 * ((rho >> 2) & 1) + 4 * (rho >> 3) + (rho & 3); }

__device__ __forceinline__ void silu16(float16v& v) {
#pragma clang fp contract(off)
  float16v t;
#pragma unroll
  for (int j = 0; j < 16; ++j) t[j] = v[j] * -1.4426950408889634f;
#pragma unroll
  for (int j = 0; j < 16; ++j) t[j] = __builtin_amdgcn_exp2f(t[j]);
#pragma unroll
  for (int j = 0; j < 16; ++j) t[j] = 1.0f + t[j];
#pragma unroll
  for (int j = 0; j < 16; ++j) t[j] = __builtin_amdgcn_rcpf(t[j]);
#pragma unroll
  for (int j = 0; j < 16; ++j) v[j] = v[j] * t[j];
}

template <int K, int CB, int PB = 2>
struct W1 {
  static constexpr int KS = K / 16;                    // K slices
  static constexpr int NCH = K / 8;                    // 16-byte chunks per pixel row
  static constexpr int RB = 2 * K;                     // bytes per pixel row
  static constexpr int TP = 32 * PB * (8 / CB);        // pixels per tile
  static constexpr int WPX = 32 * PB;                  // pixels per wave
  static constexpr int TILE_BYTES = TP * RB;
  static constexpr int NPIECES = TILE_BYTES / 1024;
  static constexpr int P_IT = NPIECES / NWAVES;
  static constexpr int NBUF = (3 * TILE_BYTES <= 120 * 1024) ? 3 : 2;
  static constexpr int SWM = (K % 128 == 0) ? 15 : 7;  // XOR mask on the chunk index
  // XOR key of a pixel row: 256-byte multiples (K % 128 == 0) start every row on bank 0 -> pixel & 15; 384-byte rows (K = 192)
  // alternate between two bank halves -> (pixel >> 1) & 7 (pixel & 7 left a 2-way conflict: 9 % of the K = 192 launch by
  // SQ_LDS_BANK_CONFLICT, profiles/r03_pmc_mfma.json before the fix)
  static constexpr int SWS = (K % 128 == 0) ? 0 : 1;
  static constexpr int STG_OFF = NBUF * TILE_BYTES;    // 8 waves x 32 pixels x 64 bytes (one pixel block at a time)
  static constexpr int BIAS_OFF = STG_OFF + NWAVES * 2048;
  static constexpr int LDS_BYTES = BIAS_OFF + 1024;
  static_assert(K % 64 == 0 && K <= 512 && (CB == 4 || CB == 8) && (PB == 1 || PB == 2) && NPIECES % NWAVES == 0, "shape");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS");
};

__device__ __forceinline__ void fdivmod(int n, int d, float inv_d, int& q, int& r) {   // 0 <= n < 2^24 (as in conv_igemm.hip)
  q = (int)((float)n * inv_d);
  r = n - q * d;
  if (r < 0) { r += d; --q; }
  if (r >= d) { r -= d; ++q; }
}

template <int K, int CB, int PB, bool SPLIT>
__global__ __launch_bounds__(512, 2) void conv1x1_wreg_kernel(const ConvArgs a, int ntiles_px, int ntiles) {
  using C = W1<K, CB, PB>;
  static_assert(!SPLIT || (K == 384 && CB == 4 && PB == 1 && C::SWM == 15 && C::SWS == 0), "the split form: 256 read-through + 128 own channels");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nwg = gridDim.x;
  const int n = lane & 31, h = lane >> 5;
  const int m = CB == 8 ? wave : (wave & 3);          // channel block inside the channel tile
  const int ph = CB == 8 ? 0 : (wave >> 2);           // pixel half of the tile

  // virtual block vb -> (channel tile, pixel tile): the pixel tiles of one channel tile are contiguous in vb, so a block's
  // walk (vb += gridDim.x) changes its channel tile at most once in a launch (Cout = 512: two channel tiles)
  auto tile_of = [&](int vb, int& ct, int& p0) __attribute__((always_inline)) {
    ct = vb / ntiles_px;
    p0 = (vb - ct * ntiles_px) * C::TP;
  };
  int ct_cur = -1;
  half8 wv[C::KS];
  auto load_weights = [&](int ct) __attribute__((always_inline)) {
    if (a.wf) {    // fragment-ordered copy [channel block][slice]: one coalesced 1 KiB load per fragment
      const half_t* wp = a.wf + (long)(ct * CB + m) * C::KS * 512 + lane * 8;
#pragma unroll
      for (int s = 0; s < C::KS; ++s) wv[s] = *(const half8*)(wp + 512 * s);
    } else {
      const half_t* wp = a.w + (long)(ct * CB * 32 + 32 * m + row_plain(n)) * a.Kpad + 8 * h;
#pragma unroll
      for (int s = 0; s < C::KS; ++s) wv[s] = *(const half8*)(wp + 16 * s);
    }
    if (tid < CB * 32) ((float*)(smem + C::BIAS_OFF))[tid] = a.bias[ct * CB * 32 + tid];
    ct_cur = ct;
  };

  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)((long)a.M * a.ldx * 2), 0x00020000);
  const int HW = a.Ho * a.Wo;
  const __amdgpu_buffer_rsrc_t rs_x2 = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(SPLIT ? a.x2 : a.x), 0, SPLIT ? (int)((long)(a.M / HW) * a.x2_bstride * 2) : 0, 0x00020000);
  const float inv_hw = 1.0f / (float)HW, inv_w = 1.0f / (float)a.Wo;
  // ---- activation tile pieces: wave w owns pieces g = w + 8 i; lane-linear chunk id c = 64 g + lane = (pixel, slot)
  // (the per-piece offsets are rebuilt from mbcnt at every issue: held in registers across the K loop they are spilled in the
  // K = 512 form, and a scratch reload waits on vmcnt(0), i.e. on the stores of the tile just finished)
  auto issue_tile = [&](int p0, int buf) __attribute__((always_inline)) {
    const int soff = p0 * a.ldx * 2;
    const int npx = a.M - p0;                          // pixels of this tile inside the tensor (< TP only in the last pixel tile)
    int ln;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
#pragma unroll
    for (int i = 0; i < C::P_IT; ++i) {
      const int cid = 64 * (wave + NWAVES * i) + ln;
      const int px = cid / C::NCH, slot = cid - px * C::NCH;
      const int cc = slot ^ ((px >> C::SWS) & C::SWM);
      char* const dst = smem + buf * C::TILE_BYTES + (wave + NWAVES * i) * 1024;
      if constexpr (SPLIT) {
        // the tile as TWO images, [pixel][256 low-resolution channels] then [pixel][128 channels of x]: pieces 0 .. 31 belong to the
        // first, 32 .. 47 to the second, so a piece has ONE source (a piece of the interleaved row format holds chunks of both: two
        // masked instructions per piece made the launch DMA-issue-bound, 49.6 us)
        const int g = wave + NWAVES * i;               // (i < 4: low image, compile-time per i)
        if (i < 4) {
          const int c2 = 64 * g + ln, px2 = c2 >> 5, slot2 = c2 & 31;
          const int cc2 = (slot2 & ~15) | ((slot2 ^ px2) & 15);
          int b, pix, ho, wo;
          fdivmod(p0 + (px2 < npx ? px2 : 0), HW, inv_hw, b, pix);
          fdivmod(pix, a.Wo, inv_w, ho, wo);
          dma16(rs_x2, px2 < npx ? (int)(((long)b * a.x2_bstride + ((long)(ho >> 1) * (a.Wo >> 1) + (wo >> 1)) * a.ldx2 + cc2 * 8) * 2)
                                 : (int)0x80000000, 0, dst);
        } else {
          const int c2 = 64 * (g - 32) + ln, px2 = c2 >> 4, slot2 = c2 & 15;
          dma16(rs_x, px2 < npx ? (px2 * a.ldx + 256 + ((slot2 ^ px2) & 15) * 8) * 2 : (int)0x80000000, soff, dst);
        }
      } else {
        dma16(rs_x, px < npx ? (px * a.ldx + cc * 8) * 2 : (int)0x80000000, soff, dst);   // past the tensor: zeros
      }
    }
  };
  // ---- fragment offsets: pixel (WPX ph + 32 pb + n) of the tile, chunk 2 s + h
  int offp[PB], swz[PB];
#pragma unroll
  for (int pb = 0; pb < PB; ++pb) {
    const int px = C::WPX * ph + 32 * pb + n;
    offp[pb] = SPLIT ? px * 512 : px * C::RB;            // (SPLIT: offset in the low image; the other image: TP * 512 + px * 256)
    swz[pb] = (px >> C::SWS) & C::SWM;
  }
  // output staging: this wave's 64 pixels x 32 channels (64-byte rows); chunk XOR (pixel >> 1) & 3
  char* const stg = smem + C::STG_OFF + wave * 2048;
  const int st_p = lane >> 2, st_k = lane & 3;

  int vb = blockIdx.x, ct, p0, nct = 0, np0 = 0;
  tile_of(vb, ct, p0);
  load_weights(ct);
  issue_tile(p0, 0);
  bool have1 = vb + nwg < ntiles;
  if (have1) {
    tile_of(vb + nwg, nct, np0);
    issue_tile(np0, 1);
  }
  __builtin_amdgcn_s_waitcnt(0x0070);                    // (the builtin: the compiler then does not re-wait for the weights in the loop)
  __builtin_amdgcn_s_barrier();

  for (int it = 0;; ++it) {
    // with three buffers the tile after next is issued now; with two, the next tile is issued after this tile's reads
    int nnct = 0, nnp0 = 0;
    bool have2 = false;
    if (C::NBUF == 3) {
      have2 = have1 && vb + 2 * nwg < ntiles;
      if (have2) {
        tile_of(vb + 2 * nwg, nnct, nnp0);
        issue_tile(nnp0, (it + 2) % 3);
      }
    }
    const char* const xb = smem + (it % C::NBUF) * C::TILE_BYTES;
    float16v acc[PB];
    {
      float16v bv;
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const float4v u = *(const float4v*)(smem + C::BIAS_OFF + (32 * m + 16 * h + 4 * qd) * 4);
        bv[qd * 4 + 0] = u[0]; bv[qd * 4 + 1] = u[1]; bv[qd * 4 + 2] = u[2]; bv[qd * 4 + 3] = u[3];
      }
#pragma unroll
      for (int pb = 0; pb < PB; ++pb) acc[pb] = bv;
    }
    // ---- K loop in groups of four slices; the fragments of group g + 1 are read under the MFMAs of group g
    half8 fr[2][PB][4];
    auto read_group = [&](int g, int set) __attribute__((always_inline)) {
#pragma unroll
      for (int pb = 0; pb < PB; ++pb)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const int ch = 2 * (4 * g + s) + h;
          if (SPLIT && 4 * g + s >= 16)      // channels 256 .. 383: the second image (256-byte rows)
            fr[set][pb][s] = *(const half8*)(xb + C::TP * 512 + (offp[pb] >> 1) + ((((ch - 32) ^ swz[pb]) & 15) << 4));
          else
            fr[set][pb][s] = *(const half8*)(xb + offp[pb] + (((ch & ~C::SWM) | ((ch ^ swz[pb]) & C::SWM)) << 4));
        }
    };
    read_group(0, 0);
#pragma unroll
    for (int g = 0; g < C::KS / 4; ++g) {
      if (g + 1 < C::KS / 4) read_group(g + 1, (g + 1) & 1);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int pb = 0; pb < PB; ++pb)
          acc[pb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wv[4 * g + s], fr[g & 1][pb][s], acc[pb], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- epilogue, one pixel block at a time: SiLU, fp16, transpose through 2 KB of LDS, 64-byte row segments out
    {
      half_t* const yb = (half_t*)a.y + (long)(p0 + C::WPX * ph) * a.ldy + ct * CB * 32 + 32 * m;
#pragma unroll
      for (int pb = 0; pb < PB; ++pb) {
        if (a.act) silu16(acc[pb]);
        half8 o0, o1;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          o0[j] = m355_to_half(acc[pb][j]);
          o1[j] = m355_to_half(acc[pb][8 + j]);
        }
        if (pb) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // (the reads of block 0 have returned)
        *(half8*)(stg + n * 64 + (((2 * h) ^ ((n >> 1) & 3)) << 4)) = o0;
        *(half8*)(stg + n * 64 + (((2 * h + 1) ^ ((n >> 1) & 3)) << 4)) = o1;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int p = 16 * i + st_p;
          const half8 v = *(const half8*)(stg + p * 64 + ((st_k ^ ((p >> 1) & 3)) << 4));
          if (p0 + C::WPX * ph + 32 * pb + p < a.M) *(half8*)(yb + (long)(32 * pb + p) * a.ldy + st_k * 8) = v;
        }
      }
    }
    if (!have1) break;
    if (C::NBUF == 3) {
      // the next tile (issued one iteration ago, or in the prologue) has landed for this wave: everything older than this
      // iteration's own pieces and its four stores
      // (a partial last pixel tile may have issued fewer than four stores: its wait covers them too)
      const bool partial = p0 + C::TP > a.M;
      if (partial) {
        if (have2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(C::P_IT) : "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      } else if (have2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(C::P_IT + 2 * PB) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * PB) : "memory");
      __builtin_amdgcn_s_barrier();
    } else {
      // two buffers: every wave is done reading this tile -> issue the tile after next into it, then wait for the next tile
      const bool h2 = vb + 2 * nwg < ntiles;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (h2) {
        tile_of(vb + 2 * nwg, nnct, nnp0);
        issue_tile(nnp0, it % 2);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::P_IT) : "memory");     // all but the pieces just issued
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      have2 = h2;
      __builtin_amdgcn_s_barrier();
    }
    vb += nwg;
    ct = nct; p0 = np0;
    nct = nnct; np0 = nnp0;
    have1 = have2;
    if (ct != ct_cur) {                                   // (Cout = 512: the walk crossed into the second channel tile)
      load_weights(ct);
      __builtin_amdgcn_s_waitcnt(0x0070);
      __builtin_amdgcn_s_barrier();
    }
  }
}

template <int K, int CB, int PB = 2, bool SPLIT = false>
int launch_w1(const ConvArgs& a, hipStream_t s) {
  using C = W1<K, CB, PB>;
  const int ntiles_px = (a.M + C::TP - 1) / C::TP, ctiles = a.Cout / (CB * 32);   // (the last pixel tile may be partial)
  const int ntiles = ntiles_px * ctiles;
  static int slots = 0;
  auto k = conv1x1_wreg_kernel<K, CB, PB, SPLIT>;
  if (!slots) {
    hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
      return -2;
    slots = cus < 8 ? 8 : cus;
  }
  // a block must not change its channel tile in the middle of a uniform loop more than once and all blocks must agree on
  // the barrier count of that change: every block's walk crosses at most one boundary, each crossing is block-local
  const int grid = ntiles <= slots ? ntiles : slots;
  hipLaunchKernelGGL(k, dim3(grid), dim3(64 * NWAVES), C::LDS_BYTES, s, a, ntiles_px, ntiles);
  return (int)hipGetLastError();
}

}  // namespace

// Eligibility: 1x1 / s1, fp16 in and out, no residual / fused epilogue, dense pixel rows (the batch is one flat pixel axis), K in
// {128, 192, 256, 384, 512}, Cout a multiple of 128; any pixel count (a partial last pixel tile is masked).  The upsample
// read-through (csplit) only in the 384 -> 128 form.
bool conv1x1_wreg_ok(const ConvArgs& a) {
  if (a.ksize != 1 || a.stride != 1 || a.pad != 0 || a.out_f32 || a.convt_co > 0 || a.tmode || a.phase || a.w2 || a.dec_preds ||
      a.res)
    return false;
  if (a.Cin != 128 && a.Cin != 192 && a.Cin != 256 && a.Cin != 384 && a.Cin != 512) return false;
  if (a.Cout % 128 || a.Cout > 512 || a.ldx % 8 || a.ldy % 8 || a.Kpad < a.Cin || a.Kpad % 8) return false;
  if (a.Ho != a.Hi || a.Wo != a.Wi) return false;
  if (a.x_bstride != (long)a.Hi * a.Wi * a.ldx || a.y_bstride != (long)a.Ho * a.Wo * a.ldy) return false;   // flat pixel axis
  const int cb = a.Cout % 256 == 0 ? 8 : 4;
  if (cb == 4 && a.Cin > 384) return false;              // (a 64-pixel tile of K = 512 leaves the 128-channel form no second buffer worth having)
  if (a.csplit > 0) {
    // upsample read-through: the 384 -> 128 instance only; the low-resolution part in whole 16-chunk groups (the piece accounting of
    // the kernel), even maps, every pixel of x2 reachable through one buffer descriptor, pixel count within the float divide
    if (!(a.Cin == 384 && cb == 4) || !a.x2 || a.csplit != 256 || a.ldx2 % 8 || a.x2_bstride % 8 || (a.Hi & 1) ||
        (a.Wi & 1) || a.M >= (1 << 24) || a.M % ((long)a.Ho * a.Wo))
      return false;
    if ((a.M / ((long)a.Ho * a.Wo)) * a.x2_bstride * 2 >= (1L << 31)) return false;
  }
  return (long)a.M * a.ldx * 2 < (1L << 31) && (long)a.M * a.ldy < (1L << 31);
}

int launch_conv1x1_wreg(const ConvArgs& a, hipStream_t s) {
  if (!conv1x1_wreg_ok(a) || !conv_rows_covered(a, 128)) return -1;
  const bool c8 = a.Cout % 256 == 0;
  if (a.csplit > 0 && !(a.Cin == 384 && !c8)) return -1;
  switch (a.Cin) {
    case 128: return c8 ? launch_w1<128, 8>(a, s) : launch_w1<128, 4>(a, s);
    case 192: return c8 ? launch_w1<192, 8>(a, s) : launch_w1<192, 4>(a, s);
    case 256: return c8 ? launch_w1<256, 8>(a, s) : launch_w1<256, 4>(a, s);
    case 384: return c8 ? launch_w1<384, 8>(a, s) : (a.csplit > 0 ? launch_w1<384, 4, 1, true>(a, s) : launch_w1<384, 4, 1, false>(a, s));
    case 512: return c8 ? launch_w1<512, 8>(a, s) : -1;
  }
  return -1;
}

}  // namespace m355
