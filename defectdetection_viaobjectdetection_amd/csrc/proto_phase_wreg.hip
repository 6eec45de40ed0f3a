// The composed Proto block -- ConvTranspose2d(128, 128, 2, 2) -> Conv3x3(128 -> 128) + BN + SiLU -> Conv1x1(128 -> 32) + BN + SiLU
// -- as a persistent kernel with the WEIGHTS IN REGISTERS (gfx950, v_mfma_f32_32x32x16_f16).
//
// Replaces: upstream's Proto.upsample / Proto.cv2 / Proto.cv3 (nn.modules.block.Proto), reached through
// /root/reference/BscanBased/yolo8_seg_predict.py:8.  The algebra is the engine's composed form (engine.hip, DESIGN.md
// section 4): the transposed convolution has no activation, so upsample -> 3x3 is, per output phase (py, px), a 2x2
// convolution over the LOW-resolution map with host-composed weights [4 phases][128][(a, b, cin)] and a bias table by
// border class of the output pixel.
//
// Why its own kernel.  On the im2col kernel this was the most expensive launch of the forward (183-205 us for 114 GFLOP,
// 0.24 of the MFMA peak, 409 600 LDS bank conflicts): every 128-pixel tile re-streamed 128 KB of weights through LDS-DMA and
// gathered its taps pixel by pixel.  Here a block owns ONE PHASE for its whole life: wave (m, half) keeps the 32 fragments
// of channel block m of that phase in 128 VGPRs, the 9 x 17-pixel patch of a tile (8 x 16 low-resolution pixels) is staged
// once by LDS-DMA and the four taps read shifted windows of it; the 128-channel result goes to LDS as fp16 and the 1x1
// (its eight weight fragments in LDS) runs on it; the 32-channel prototype rows leave as whole 64-byte pixel rows.
//
// Block = 8 waves, one block per CU, phase = blockIdx.x & 3, tiles walked with stride gridDim.x / 4.  Wave (m = wave & 3,
// half = wave >> 2): channels 32 m .. 32 m + 31 x tile rows 4 half .. 4 half + 3 (two 32-pixel MFMA blocks of 2 rows x 16).
// LDS patch: one 256-byte row per pixel, pitch 17, 16-byte chunk index XOR-ed with (patch column & 15): the 16 lanes of a
// ds_read_b128 service group cover 16 consecutive patch columns (of two rows) -> 16 distinct bank groups for every tap.
#include <stdio.h>
#include <stdlib.h>

#include <hip/hip_runtime.h>

#include "common.h"

namespace m355 {
namespace {

typedef float float16v __attribute__((ext_vector_type(16)));

constexpr int TH = 8, TW = 16, PP = 17, ROWB = 256, NWAVES = 8;
constexpr int PROWS = (TH + 1) * PP;                 // 153 patch pixels
constexpr int NPIECES = (PROWS + 3) / 4;             // 39 DMA pieces of 4 rows
constexpr int G_SPLIT = 27, P_IT = 7;                // DMA pieces of waves 0-3 / per wave (issue_patch)
constexpr int PATCH_BYTES = NPIECES * 1024;          // 39936
constexpr int NBUF = 2;
constexpr int Z_OFF = NBUF * PATCH_BYTES;            // 128 pixels x 256 bytes: SiLU(phase conv) as fp16
constexpr int W3_OFF = Z_OFF + 128 * 256;            // eight fragments of the 1x1 (lane-linear)
constexpr int BIAS_OFF = W3_OFF + 8 * 1024;          // [9][128] floats + 32 floats
constexpr int STG_OFF = BIAS_OFF + 9 * 512 + 128;    // output staging: 4 waves x 32 pixels x 64 bytes
constexpr int TAB_OFF = STG_OFF + 4 * 2048;          // DMA offset table: [39 pieces][64 lanes] ints
constexpr int LDS_BYTES = TAB_OFF + 39 * 256;        // 143744

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

__global__ __launch_bounds__(512, 2) void proto_phase_wreg_kernel(const ConvArgs a, int tiles_x, int tiles_y, int ntiles, int sx, int sy,
                                                                 int sb, int prio, unsigned long long* stamps) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = a.Hi, W = a.Wi;                       // low-resolution map
  const int n = lane & 31, h = lane >> 5;
  const int m = wave & 3, half = wave >> 2;
  // Block -> (phase, tile stream).  Workgroups go round-robin over the 8 XCDs, so with a grid that is a multiple of 32 the four
  // phases of tile stream vb are the blocks vb % 8 + 8 (q + 4 (vb / 8)): same XCD, the shared input patch is fetched into ONE L2
  // (with phase = blockIdx & 3 every XCD pulled the whole input through its own L2).
  const int nblk = gridDim.x >> 2;                    // blocks per phase = tile streams
  const bool xcd_map = (gridDim.x & 31) == 0;
  const int q = xcd_map ? (blockIdx.x >> 3) & 3 : blockIdx.x & 3, py = q >> 1, px = q & 1;
  const int vb = xcd_map ? (blockIdx.x & 7) + 8 * (blockIdx.x >> 5) : blockIdx.x >> 2;

  // bias table [9][128] + the 1x1's 32 biases + the 1x1's fragments -> LDS; this wave's phase-conv fragments -> registers
  for (int i = tid; i < 9 * 128; i += 64 * NWAVES) ((float*)(smem + BIAS_OFF))[i] = a.bias[i];
  if (tid < 32) ((float*)(smem + BIAS_OFF + 9 * 512))[tid] = a.bias2[tid];
  for (int i = tid; i < 8 * 64; i += 64 * NWAVES) *(float4v*)(smem + W3_OFF + i * 16) = *(const float4v*)(a.wf2 + (long)i * 8);
  half8 wv[32];
  {
    const half_t* wp = a.wf + (long)(q * 4 + m) * 32 * 512 + lane * 8;
#pragma unroll
    for (int s = 0; s < 32; ++s) wv[s] = *(const half8*)(wp + 512 * s);
  }

  const int nimg = a.M / (a.Ho * a.Wo);
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
      (void*)a.x, 0, (int)((nimg - 1) * a.x_bstride + (long)H * W * a.ldx) * 2, 0x00020000);
  const int img_stride = (int)a.x_bstride * 2;
  // ---- patch pieces: wave w owns pieces g = w + 8 i (4 LDS rows each); lane = (row 4 g + lane / 16, chunk slot lane % 16).
  // Patch pixel (pr, pc) = input pixel (y0 - 1 + py + pr, x0 - 1 + px + pc).
  // (row / column / offset of each piece are recomputed per issue: five live registers here were spilled, and a scratch reload
  // waits on vmcnt, i.e. on the DMA issued just before it -- the issue loop ran at one memory latency per piece)
  auto tile_of = [&](int t, int& tb, int& y0, int& x0) __attribute__((always_inline)) {
    const int tx = t % tiles_x;
    const int rest = t / tiles_x;
    tb = rest / tiles_y;
    y0 = (rest - tb * tiles_y) * TH;
    x0 = tx * TW;
  };
  // DMA offset table (tile independent): entry (piece g, lane l) = byte offset of that lane's 16 bytes from the patch origin,
  // | 1 top patch row | 2 bottom row | 4 left column | 8 right column (low four bits: offsets are multiples of 16); -1 = lane
  // past the patch.  Computed per issue, this was ~25 VALU instructions per piece (a division by 17 and two multiplies)
  // beside the partner wave's K loop; from the table it is a ds_read and five.
  for (int i = tid; i < NPIECES * 64; i += 64 * NWAVES) {
    const int g = i >> 6, l = i & 63;
    const int R = 4 * g + (l >> 4);
    const int pr = R / PP, pc = R - pr * PP;
    int e = -1;
    if (R < PROWS)
      e = (((pr * W + pc) * a.ldx + (((l ^ pc) & 15) << 3)) * 2) | (pr == 0 ? 1 : 0) | (pr == TH ? 2 : 0) | (pc == 0 ? 4 : 0) | (pc == TW ? 8 : 0);
    ((int*)(smem + TAB_OFF))[i] = e;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                        // (the table is read by the first issue below)
  auto issue_patch = [&](int tb, int y0, int x0, int buf) __attribute__((always_inline)) {
    const int oy = y0 - 1 + py, ox = x0 - 1 + px;
    const int origin = ((oy * W + ox) * a.ldx) * 2;
    // which border lines of the patch leave the map for this (tile, phase); bit 31 rejects the lanes past the patch
    const int reject = (int)0x80000000 | (oy < 0 ? 1 : 0) | (oy + TH >= H ? 2 : 0) | (ox < 0 ? 4 : 0) | (ox + TW >= W ? 8 : 0);
    int ln;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
    // Waves 0-3 (whose next stop is barrier A, where they wait for the other team's K loop) take 27 of the 39 pieces, waves
    // 4-7 (which go straight into their K loop: the critical path of the tile) the other 12.  Wave w of a team: g = g0 + w + 4 i.
    const int g0 = wave < 4 ? 0 : G_SPLIT, g1 = wave < 4 ? G_SPLIT : NPIECES;
    const int* const tab = (const int*)(smem + TAB_OFF) + (g0 + m) * 64 + ln;
    int e[P_IT];
#pragma unroll
    for (int i = 0; i < P_IT; ++i) e[i] = (g0 + m + 4 * i < g1) ? tab[256 * i] : -1;
#pragma unroll
    for (int i = 0; i < P_IT; ++i) {
      const int g = g0 + m + 4 * i;
      if (g < g1)
        dma16(rs_x, (e[i] & reject) == 0 ? origin + (e[i] & ~15) : (int)0x80000000, tb * img_stride,
              smem + buf * PATCH_BYTES + g * 1024);   // out of range = zeros
    }
  };
  // ---- fragment offsets: pixel block pb = tile rows 4 half + 2 pb, + (n >> 4); column n & 15; tap (ta, tb) adds (ta, tb)
  const int prow = 4 * half + (n >> 4), pcol = n & 15;
  int offp[2][2];                                      // [pb][tap column]: byte offset of the patch row, chunk swizzle of the column
#pragma unroll
  for (int pb = 0; pb < 2; ++pb)
#pragma unroll
    for (int tc = 0; tc < 2; ++tc) offp[pb][tc] = ((prow + 2 * pb) * PP + pcol + tc) * ROWB;
  const int swc0 = pcol & 15, swc1 = (pcol + 1) & 15;

  int t = vb, tb, y0, x0, ntb = 0, ny0 = 0, nx0 = 0;
  tile_of(t, tb, y0, x0);
  // the walk in tile units, stepped with carries from here on (the scalar divisions of tile_of cost ~1 k cycles per tile)
  int wb = tb, wy = y0 / TH, wx = x0 / TW;
  issue_patch(tb, y0, x0, 0);
  __builtin_amdgcn_s_waitcnt(0x0070);                  // (the builtin: the compiler does not re-wait for the weight loads in the loop)
  __builtin_amdgcn_s_barrier();

  // diagnostic launches only (M355_PROTOR_STAMPS): cycles per section and wave
  unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = stamps ? __builtin_amdgcn_s_memtime() : 0;
  int ntile = 0;
#define PP_STAMP(k)                                                                                       \
  if (stamps) {                                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                                    \
    const unsigned long long tn = __builtin_amdgcn_s_memtime();                                           \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                                    \
    tacc[k] += tn - tlast;                                                                                \
    tlast = tn;                                                                                           \
  }
  // proto.cv3 of one finished tile: 32 channels x 128 pixels x K = 128 from Z; waves 0-3 take one 32-pixel block each
  auto cv3_tile = [&](int tb, int y0, int x0) __attribute__((always_inline)) {
    // (lane-derived values are rebuilt from mbcnt here: kept live across the phase conv they are spilled, and a scratch reload
    // waits on vmcnt, i.e. on the patch DMA issued just before)
    int ln;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
    const int n = ln & 31, h = ln >> 5, st_p = ln >> 2, st_k = ln & 3;
    char* const stg = smem + STG_OFF + wave * 2048;
    float16v o;
    {
      const float* b2 = (const float*)(smem + BIAS_OFF + 9 * 512) + 16 * h;
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const float4v u = *(const float4v*)(b2 + 4 * qd);
        o[qd * 4 + 0] = u[0]; o[qd * 4 + 1] = u[1]; o[qd * 4 + 2] = u[2]; o[qd * 4 + 3] = u[3];
      }
    }
    const int zp = 32 * wave + n;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const half8 wf3 = *(const half8*)(smem + W3_OFF + s * 1024 + ln * 16);
      const half8 zf = *(const half8*)(smem + Z_OFF + zp * 256 + (((2 * s + h) ^ (zp & 15)) << 4));
      o = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf3, zf, o, 0, 0, 0);
    }
    silu16(o);
    half8 o0, o1;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      o0[j] = m355_to_half(o[j]);
      o1[j] = m355_to_half(o[8 + j]);
    }
    // transpose through LDS: lane (pixel, half) -> four lanes per 64-byte pixel row; chunk XOR (pixel >> 1) & 3
    *(half8*)(stg + n * 64 + (((2 * h) ^ ((n >> 1) & 3)) << 4)) = o0;
    *(half8*)(stg + n * 64 + (((2 * h + 1) ^ ((n >> 1) & 3)) << 4)) = o1;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // staged pixel p of block `wave` = tile pixel 32 wave + p = (row 2 wave + (p >> 4), column p & 15)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int p = 16 * i + st_p;
      const half8 v = *(const half8*)(stg + p * 64 + ((st_k ^ ((p >> 1) & 3)) << 4));
      const int Y = 2 * (y0 + 2 * wave + i) + py, X = 2 * (x0 + st_p) + px;
      *(half8*)((half_t*)a.y + (long)tb * a.y_bstride + ((long)Y * (2 * W) + X) * a.ldy + st_k * 8) = v;
    }
  };
  // Schedule (per tile, two barriers): the two waves of a SIMD run half a tile apart.  Waves 4-7 start the phase conv of tile i
  // at once (MFMA pipe to themselves) while waves 0-3 finish tile i - 1 (proto.cv3 from Z, SiLU, stores); barrier A; waves 0-3
  // run their phase conv while waves 4-7 are in bias + SiLU + Z (VALU beside the partner's MFMAs); barrier B = Z(i) complete
  // and patch(i + 1) landed.  Z(i - 1) is read before barrier A and Z(i) written after it: one Z buffer.
  int ptb = 0, py0 = 0, px0 = 0;
  for (int it = 0;; ++it) {
    ++ntile;
    const bool more = t + nblk < ntiles;
    if (more) {                                        // the next tile's patch streams in under this tile's arithmetic
      wx += sx;
      if (wx >= tiles_x) { wx -= tiles_x; ++wy; }
      wy += sy;
      if (wy >= tiles_y) { wy -= tiles_y; ++wb; }
      wb += sb;
      ntb = wb; ny0 = wy * TH; nx0 = wx * TW;
      issue_patch(ntb, ny0, nx0, (it + 1) & 1);
    }
    PP_STAMP(0)   // tile decode + DMA issue
    if (wave < 4) {
      if (it > 0) cv3_tile(ptb, py0, px0);
      PP_STAMP(1)   // proto.cv3 + stores of the previous tile (waves 0-3)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                    // A (waves 0-3)
    }
    PP_STAMP(2)   // barrier A, waves 0-3
    const char* const pbuf = smem + (it & 1) * PATCH_BYTES;
    // ---- phase conv: 4 taps x 8 slices x 2 pixel blocks; the fragments of the next half tap are read under the MFMAs
    float16v acc[2];
    acc[0] = (float16v)0.f;
    acc[1] = (float16v)0.f;
    // fragments: one set per pixel block; the set of (half tap + 1, block) is read right behind the four MFMAs of
    // (half tap, block), i.e. four MFMAs (128 cycles) ahead of its use
    if (prio) __builtin_amdgcn_s_setprio(1);
    half8 fr[2][4];
    auto read_half = [&](int ht, int pb) __attribute__((always_inline)) {   // ht = tap * 2 + channel half (64 channels = 4 slices)
      const int tap = ht >> 1, ta = tap >> 1, tc = tap & 1, c0 = (ht & 1) * 8;
#pragma unroll
      for (int s = 0; s < 4; ++s)
        fr[pb][s] = *(const half8*)(pbuf + offp[pb][tc] + ta * PP * ROWB + (((c0 + 2 * s + h) ^ (tc ? swc1 : swc0)) << 4));
    };
    read_half(0, 0);
    read_half(0, 1);
#pragma unroll
    for (int ht = 0; ht < 8; ++ht) {
#pragma unroll
      for (int pb = 0; pb < 2; ++pb) {
#pragma unroll
        for (int s = 0; s < 4; ++s) acc[pb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wv[4 * ht + s], fr[pb][s], acc[pb], 0, 0, 0);
        if (ht < 7) read_half(ht + 1, pb);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (prio) __builtin_amdgcn_s_setprio(0);
    PP_STAMP(3)   // reads + MFMAs
    if (wave >= 4) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                    // A (waves 4-7): Z(i - 1) has been consumed
    }
    PP_STAMP(4)   // barrier A, waves 4-7
    // ---- Z = SiLU(acc + bias[border class of the OUTPUT pixel]) -> fp16 -> LDS [pixel][128 ch], chunk XOR (pixel & 15)
#pragma unroll
    for (int pb = 0; pb < 2; ++pb) {
      const int hh = y0 + prow + 2 * pb, ww = x0 + pcol;            // low-resolution pixel of this lane
      const int Y = 2 * hh + py, X = 2 * ww + px;
      const int ry = Y == 0 ? 0 : (Y == 2 * H - 1 ? 2 : 1), rx = X == 0 ? 0 : (X == 2 * W - 1 ? 2 : 1);
      const float* bp = (const float*)(smem + BIAS_OFF) + (ry * 3 + rx) * 128 + 32 * m + 16 * h;
      {
#pragma clang fp contract(off)
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          const float4v u = *(const float4v*)(bp + 4 * qd);
          acc[pb][qd * 4 + 0] = acc[pb][qd * 4 + 0] + u[0]; acc[pb][qd * 4 + 1] = acc[pb][qd * 4 + 1] + u[1];
          acc[pb][qd * 4 + 2] = acc[pb][qd * 4 + 2] + u[2]; acc[pb][qd * 4 + 3] = acc[pb][qd * 4 + 3] + u[3];
        }
      }
      if (a.act) silu16(acc[pb]);
      half8 o0, o1;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        o0[j] = m355_to_half(acc[pb][j]);
        o1[j] = m355_to_half(acc[pb][8 + j]);
      }
      const int zp = 64 * half + 32 * pb + n;                        // tile pixel index = row * 16 + column
      const int c = 4 * m + 2 * h;                                   // chunk of channels 32 m + 16 h ..
      *(half8*)(smem + Z_OFF + zp * 256 + ((c ^ (zp & 15)) << 4)) = o0;
      *(half8*)(smem + Z_OFF + zp * 256 + (((c + 1) ^ (zp & 15)) << 4)) = o1;
    }
    PP_STAMP(5)   // bias + SiLU + Z writes
    // the next patch has landed for this wave (waves 0-3: their stores were issued before it), Z is written
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                      // B
    PP_STAMP(6)   // wait patch + barrier B
    ptb = tb; py0 = y0; px0 = x0;
    if (!more) break;
    t += nblk;
    tb = ntb; y0 = ny0; x0 = nx0;
  }
  if (wave < 4) cv3_tile(ptb, py0, px0);               // the last tile
  if (stamps && lane == 0) {
    unsigned long long* o = stamps + ((long)blockIdx.x * NWAVES + wave) * 8;
    for (int k = 0; k < 7; ++k) o[k] = tacc[k];
    o[7] = (unsigned long long)ntile;
  }
#undef PP_STAMP
}

}  // namespace

// Eligibility: the composed Proto launch (phase conv 128 -> 128 + 1x1 128 -> 32) with fragment-ordered weights, low-resolution
// map a multiple of the 8 x 16 tile.
bool proto_phase_wreg_ok(const ConvArgs& a) {
  if (!a.phase || a.ksize != 2 || a.convt_co != 128 || a.Cin != 128 || a.cout2 != 32 || !a.wf || !a.wf2 || !a.bias2 || !a.bias) return false;
  if (a.out_f32 || a.tmode || a.res || a.csplit || a.dec_preds || a.ldx % 8 || a.ldy != 32) return false;
  if (a.Hi % TH || a.Wi % TW || a.Ho != a.Hi || a.Wo != a.Wi) return false;
  const long nimg = a.Ho * a.Wo > 0 ? a.M / ((long)a.Ho * a.Wo) : 0;
  if (nimg < 1) return false;
  return ((nimg - 1) * a.x_bstride + (long)a.Hi * a.Wi * a.ldx) * 2 < (1L << 31);
}

int launch_proto_phase_wreg(const ConvArgs& a, hipStream_t s) {
  if (!proto_phase_wreg_ok(a)) return -1;
  const int tiles_x = a.Wi / TW, tiles_y = a.Hi / TH;
  const int B = a.M / (a.Ho * a.Wo);
  const int ntiles = B * tiles_y * tiles_x;            // per phase
  static int slots = 0;
  if (!slots) {
    hipError_t e = hipFuncSetAttribute((const void*)proto_phase_wreg_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
      return -2;
    slots = cus & ~3;
    if (slots < 4) slots = 4;
  }
  int grid = 4 * ntiles <= slots ? 4 * ntiles : slots;  // a multiple of 4: every phase gets the same number of blocks
  const int step = grid >> 2;                          // tiles between two visits of a block (= blocks per phase)
  const int sx = step % tiles_x, sy = (step / tiles_x) % tiles_y, sb = step / tiles_x / tiles_y;
  static const int prio = getenv("M355_PROTOR_PRIO") ? atoi(getenv("M355_PROTOR_PRIO")) : 0;   // experiment: s_setprio(1) around the K loop
  // diagnostic: M355_PROTOR_STAMPS=<file> -> per-wave section cycles of the LAST launch, written after a stream sync [sync]
  static const char* st_path = getenv("M355_PROTOR_STAMPS");
  static unsigned long long* d_st = nullptr;
  if (st_path && !d_st) {
    if (hipMalloc((void**)&d_st, (size_t)slots * NWAVES * 64) != hipSuccess) return -2;
    (void)hipMemset(d_st, 0, (size_t)slots * NWAVES * 64);
  }
  hipLaunchKernelGGL(proto_phase_wreg_kernel, dim3(grid), dim3(64 * NWAVES), LDS_BYTES, s, a, tiles_x, tiles_y, ntiles, sx, sy, sb, prio, d_st);
  if (st_path) {
    if (hipStreamSynchronize(s) != hipSuccess) return -2;
    const size_t nbytes = (size_t)grid * NWAVES * 64;
    unsigned long long* hbuf = (unsigned long long*)malloc(nbytes);
    (void)hipMemcpy(hbuf, d_st, nbytes, hipMemcpyDeviceToHost);
    FILE* f = fopen(st_path, "wb");
    if (f) { fwrite(hbuf, 1, nbytes, f); fclose(f); }
    free(hbuf);
  }
  return (int)hipGetLastError();
}

}  // namespace m355
