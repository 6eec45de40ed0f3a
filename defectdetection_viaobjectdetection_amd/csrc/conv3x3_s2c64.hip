// model.3 (3x3 / s2, 64 -> 128) + model.4.cv1 (1x1, 128 -> 128) of the s scale in ONE launch with both weight matrices on chip:
// registers / LDS (gfx950).  Replaces the im2col launch conv_igemm<128x128,k3+1x1> (87 us at batch 32: im2col gathers of a
// stride-2 window through L2, 425 TFLOP/s) for the Conv -> C2f.cv1 pair upstream reaches through F.conv2d (SURVEY.md A4 / A5;
// call site BscanBased/yolo8_seg_predict.py:8).  Same rounding points: the 128-channel intermediate is fp16 in LDS.
//
// Block = 8 waves (one block per CU, persistent), tile = 8 x 8 output pixels of one image = two pixel blocks of 4 rows x 8
// columns.  TEAM t = wave >> 2 owns pixel block t through both convolutions; inside a team wave m = wave & 3 owns 32
// channels: the 36 K slices of the 3x3 conv in registers (144 VGPRs, fragment-ordered copy `wf`: one coalesced 1 KiB load
// each); the 8 slices of the 1x1 conv (`wf2`) sit in LDS, lane-linear.  Per tile a wave runs
//   K      36 MFMAs 32x32x16 over the patch (one ds_read_b128 per MFMA = half the LDS read rate),
//   Z      bias (in the accumulator), SiLU, fp16 -> its team's Z image in LDS [32 pixels][128 channels],
//   S2     8 MFMAs over Z with the 1x1 weights, SiLU, fp16, transpose through LDS, 64-byte row segments out.
// The two waves of a SIMD (teams 0 and 1) run half a tile apart -- the lesson of proto_phase_wreg.hip: in lockstep their
// MFMA phases and their SiLU phases collide.  Two slots per tile, one barrier after each:
//   slot 1   team 0: S2(i - 1), K(i)            team 1: Z(i - 1)
//   slot 2   team 0: Z(i)                       team 1: S2(i - 1), K(i)
// so a team's Z image is written in one slot and read in the next, and the MFMA pipe of a SIMD serves one wave at a time.
//
// Patch: 17 x 17 input pixels x 64 channels, de-interleaved by column parity on the LDS-DMA source side -- LDS pixel
// P = patch row * 20 + position, positions 0-8 = odd input columns 2 x0 - 1 + 2 j, 9-16 = even input columns 2 x0 + 2 j,
// 17-19 unused -- so the eight output columns of a tap read eight CONSECUTIVE pixels (kw = 0: position x, kw = 1: 9 + x,
// kw = 2: x + 1).  128-byte pixel rows, chunk index XOR-ed with (P >> 1) & 7 on the source side and on the read: with the
// pitch of 20 every 16-lane ds_read_b128 service group sees 16 distinct bank groups for all nine taps (brute-forced over the
// groups of MI355X_MICROARCH.md, LDS).  The next tile's patch streams into the second buffer under this tile.
#include <stdio.h>
#include <stdlib.h>

#include <hip/hip_runtime.h>

#include "common.h"

namespace m355 {
namespace {

typedef float float16v __attribute__((ext_vector_type(16)));

constexpr int TH = 8, TW = 8, PQ = 20, PRN = 17, ROWB = 128, NWAVES = 8;
constexpr int PPIX = PRN * PQ;                       // 340 LDS pixels
constexpr int NPIECES = (PPIX + 7) / 8;              // 43 DMA pieces of 8 pixels
constexpr int PATCH_BYTES = NPIECES * 1024;          // 44032
constexpr int Z_OFF = 2 * PATCH_BYTES;               // two teams x 32 pixels x 256 bytes
constexpr int STG_OFF = Z_OFF + 2 * 8192;            // output staging: 8 waves x 32 pixels x 64 bytes
constexpr int BIAS_OFF = STG_OFF + NWAVES * 2048;    // 128 + 128 floats
constexpr int W2_OFF = BIAS_OFF + 1024;              // the 1x1's 4 x 8 fragments (lane-linear: conflict-free reads)
constexpr int LDS_BYTES = W2_OFF + 32 * 1024;        // 154624

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, int voff, int soff, char* lds) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds, 16, voff, soff, 0, 0);
}

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

__device__ __forceinline__ int lane_id() {            // volatile: lane-derived values are rebuilt where they are used, not kept
  int ln;                                             // live (= spilled) across the K loop; a scratch reload waits on vmcnt(0)
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
  return ln;
}

__global__ __launch_bounds__(512, 2) void conv3x3_s2c64_cv1_kernel(const ConvArgs a, int tiles_x, int tiles_y, int ntiles, int sx,
                                                                  int sy, int sb, int prio, unsigned long long* stamps) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = a.Hi, W = a.Wi, Wo = a.Wo, nwg = gridDim.x;
  const int m = wave & 3, team = wave >> 2;

  if (tid < 128) ((float*)(smem + BIAS_OFF))[tid] = a.bias[tid];
  else if (tid < 256) ((float*)(smem + BIAS_OFF))[tid] = a.bias2[tid - 128];

  // ---- this wave's 3x3 weights: 36 fragments in registers for the block's whole life; the 1x1's fragments in LDS (with them
  // in registers too -- 144 + 32 + 16 accumulators + 32 activation fragments -- the compiler spilled six of them and reloaded
  // them from scratch in every tile)
  half8 wv[36];
  {
    const half_t* wp = a.wf + (long)m * 36 * 512 + lane * 8;
#pragma unroll
    for (int s = 0; s < 36; ++s) wv[s] = *(const half8*)(wp + 512 * s);
  }
  for (int i = tid; i < 32 * 64; i += 64 * NWAVES) *(float4v*)(smem + W2_OFF + i * 16) = *(const float4v*)(a.wf2 + (long)i * 8);

  // ---- tile walk: virtual block vb = blockIdx.x + k * gridDim.x (gridDim.x a multiple of 8 or = ntiles); the virtual blocks
  // of one XCD cover a contiguous range of tiles
  // (decoded with divisions once; then stepped with carries -- the three scalar divisions of a decode cost ~1 k cycles per tile)
  auto tile_of = [&](int vb, int& tb, int& ty, int& tx) __attribute__((always_inline)) {
    const int xcd = vb & 7, qq = ntiles >> 3, r = ntiles & 7;
    const int L = (nwg & 7) ? vb : (xcd < r ? xcd * (qq + 1) : r * (qq + 1) + (xcd - r) * qq) + (vb >> 3);
    tx = L % tiles_x;
    const int rest = L / tiles_x;
    tb = rest / tiles_y;
    ty = rest - tb * tiles_y;
  };
  auto step_tile = [&](int& tb, int& ty, int& tx) __attribute__((always_inline)) {
    tx += sx;
    if (tx >= tiles_x) { tx -= tiles_x; ++ty; }
    ty += sy;
    if (ty >= tiles_y) { ty -= tiles_y; ++tb; }
    tb += sb;
  };
  const int nimg = a.M / (a.Ho * a.Wo);
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
      (void*)a.x, 0, (int)((nimg - 1) * a.x_bstride + (long)H * W * a.ldx) * 2, 0x00020000);
  const int img_stride = (int)a.x_bstride * 2;
  // patch pieces (8 LDS pixels each): pieces 0-15 are issued by team 1 in slot 1 and pieces 16-42 by team 0 at the top of
  // slot 2 -- the short slots (Z only; an issue costs 300-400 cycles per piece beside the partner's K loop, stamps).  With all eight waves issuing at the top of the tile the ~2 k cycles of issue sat in front of team
  // 0's S2 + K; one team issuing everything took 3.6 k cycles (11 pieces x ~25 instructions beside the partner's K loop).
  // Wave m of the team owns pieces g = g0 + m + 4 i; lane = (pixel 8 g + lane / 8, chunk slot lane % 8); 24-bit multiplies
  // (full rate; rows, columns and pixel indices are far below 2^23).
  constexpr int G_SPLIT = 16, P_IT = 7;
  auto issue_patch = [&](int tb, int y0, int x0, int buf, int g0, int g1) __attribute__((always_inline)) {
    const int ln = lane_id();
    const int p0 = 8 * (g0 + m) + (ln >> 3), slot = ln & 7;
    const int origin = (__mul24(2 * y0 - 1, W) + 2 * x0) * a.ldx * 2;   // input pixel (2 y0 - 1, 2 x0), bytes (scalar)
#pragma unroll
    for (int i = 0; i < P_IT; ++i) {
      const int g = g0 + m + 4 * i;
      if (g < g1) {
        const int P = p0 + 32 * i;
        const int pr = __umul24(P, 3277) >> 16, pos = P - pr * PQ;   // P / 20, P % 20 (exact for P < 400)
        const int dx = pos < 9 ? 2 * pos - 1 : 2 * (pos - 9);                         // column offset from 2 x0
        const bool ok = pr < PRN && pos < 17 && (unsigned)(2 * y0 - 1 + pr) < (unsigned)H && (unsigned)(2 * x0 + dx) < (unsigned)W;
        const int rel = (__mul24(__mul24(pr, W) + dx, a.ldx) + ((slot ^ ((P >> 1) & 7)) << 3)) * 2;
        dma16(rs_x, ok ? origin + rel : (int)0x80000000, tb * img_stride, smem + buf * PATCH_BYTES + g * 1024);   // out of range = zeros
      }
    }
  };

  // ---- K-loop fragment offsets of this lane: output pixel (row 4 team + n / 8, column n % 8) of the tile
  const int n = lane & 31, h = lane >> 5;
  int pk[3];                                           // LDS pixel of tap (0, kw)
  {
    const int oy = 4 * team + (n >> 3), ox = n & 7;
    pk[0] = 2 * oy * PQ + ox;
    pk[1] = 2 * oy * PQ + 9 + ox;
    pk[2] = 2 * oy * PQ + ox + 1;
  }
  char* const zt = smem + Z_OFF + team * 8192;
  char* const stg = smem + STG_OFF + wave * 2048;

  unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = stamps ? __builtin_amdgcn_s_memtime() : 0;
  int ntile = 0;
#define S2_STAMP(k)                                                                                        \
  if (stamps) {                                                                                            \
    __builtin_amdgcn_sched_barrier(0);                                                                     \
    const unsigned long long tn = __builtin_amdgcn_s_memtime();                                            \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                                     \
    tacc[k] += tn - tlast;                                                                                 \
    tlast = tn;                                                                                            \
  }

  float16v acc;
  // K: the 3x3 / s2 conv of this wave's 32 channels x 32 pixels out of patch buffer `pbuf`
  auto k_loop = [&](const char* pbuf) __attribute__((always_inline)) {
    {
      const float* bp = (const float*)(smem + BIAS_OFF) + 32 * m + 16 * h;
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const float4v u = *(const float4v*)(bp + 4 * qd);
        acc[qd * 4 + 0] = u[0]; acc[qd * 4 + 1] = u[1]; acc[qd * 4 + 2] = u[2]; acc[qd * 4 + 3] = u[3];
      }
    }
    if (prio) __builtin_amdgcn_s_setprio(1);          // (the K loop is the slot's critical path; the partner is in SiLU / stores)
    half8 fr[2][4];
    auto read_tap = [&](int tap, int b) __attribute__((always_inline)) {
      const int kh = tap / 3, kw = tap - 3 * kh;
      const int P = pk[kw] + kh * PQ;
      const char* row = pbuf + P * ROWB;
      const int e = h ^ ((P >> 1) & 7);
#pragma unroll
      for (int s = 0; s < 4; ++s) fr[b][s] = *(const half8*)(row + ((e ^ (2 * s)) << 4));
    };
    read_tap(0, 0);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      if (tap < 8) read_tap(tap + 1, (tap + 1) & 1);
#pragma unroll
      for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wv[4 * tap + s], fr[tap & 1][s], acc, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (prio) __builtin_amdgcn_s_setprio(0);
  };
  // Z: SiLU, fp16, into the team's Z image [pixel n][128 channels], chunk XOR (pixel & 15)
  auto z_write = [&]() __attribute__((always_inline)) {
    const int ln = lane_id();
    const int zn = ln & 31, zh = ln >> 5;
    if (a.act) silu16(acc);
    half8 o0, o1;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      o0[j] = m355_to_half(acc[j]);
      o1[j] = m355_to_half(acc[8 + j]);
    }
    const int c = 4 * m + 2 * zh;
    *(half8*)(zt + zn * 256 + ((c ^ (zn & 15)) << 4)) = o0;
    *(half8*)(zt + zn * 256 + (((c + 1) ^ (zn & 15)) << 4)) = o1;
  };
  // S2: the 1x1 conv of this wave's 32 output channels x its team's 32 pixels out of Z, SiLU, store
  auto stage2 = [&](int tb, int y0, int x0) __attribute__((always_inline)) {
    const int ln = lane_id();
    const int zn = ln & 31, zh = ln >> 5, st_p = ln >> 2, st_k = ln & 3;
    float16v o;
    {
      const float* b2 = (const float*)(smem + BIAS_OFF) + 128 + 32 * m + 16 * zh;
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const float4v u = *(const float4v*)(b2 + 4 * qd);
        o[qd * 4 + 0] = u[0]; o[qd * 4 + 1] = u[1]; o[qd * 4 + 2] = u[2]; o[qd * 4 + 3] = u[3];
      }
    }
    // (fragment reads two slices ahead, explicitly: left to itself the compiler read, waited and multiplied slice by slice --
    // eight LDS round trips; all sixteen at once do not fit beside the 144 weight registers)
    half8 wq[2][2], zf[2][2];
    auto s2_read = [&](int g) __attribute__((always_inline)) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        wq[g & 1][s] = *(const half8*)(smem + W2_OFF + (m * 8 + 2 * g + s) * 1024 + ln * 16);
        zf[g & 1][s] = *(const half8*)(zt + zn * 256 + (((2 * (2 * g + s) + zh) ^ (zn & 15)) << 4));
      }
    };
    s2_read(0);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      if (g < 3) s2_read(g + 1);
#pragma unroll
      for (int s = 0; s < 2; ++s) o = __builtin_amdgcn_mfma_f32_32x32x16_f16(wq[g & 1][s], zf[g & 1][s], o, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    silu16(o);
    half8 o0, o1;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      o0[j] = m355_to_half(o[j]);
      o1[j] = m355_to_half(o[8 + j]);
    }
    // transpose through LDS: lane (pixel, half) -> four lanes per 64-byte pixel row; chunk XOR (pixel >> 1) & 3
    *(half8*)(stg + zn * 64 + (((2 * zh) ^ ((zn >> 1) & 3)) << 4)) = o0;
    *(half8*)(stg + zn * 64 + (((2 * zh + 1) ^ ((zn >> 1) & 3)) << 4)) = o1;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    half_t* const yb = (half_t*)a.y + (long)tb * a.y_bstride + ((long)(y0 + 4 * team) * Wo + x0) * a.ldy + 32 * m;
#pragma unroll
    for (int i = 0; i < 2; ++i) {        // staged pixels 16 i .. 16 i + 15 = block rows 2 i, 2 i + 1 (8 columns each)
      const int p = 16 * i + st_p;
      const half8 v = *(const half8*)(stg + p * 64 + ((st_k ^ ((p >> 1) & 3)) << 4));
      *(half8*)(yb + ((long)(p >> 3) * Wo + (p & 7)) * a.ldy + st_k * 8) = v;
    }
  };

  int vb = blockIdx.x, tb, ty, tx, y0, x0, ptb = 0, py0 = 0, px0 = 0;
  tile_of(vb, tb, ty, tx);
  y0 = ty * TH; x0 = tx * TW;
  int ctb = tb;
  if (team == 1) issue_patch(ctb, y0, x0, 0, 0, G_SPLIT);
  else issue_patch(ctb, y0, x0, 0, G_SPLIT, NPIECES);
  __builtin_amdgcn_s_waitcnt(0x0070);                  // (the builtin: the compiler does not re-wait for the weight loads in the loop)
  __builtin_amdgcn_s_barrier();

  for (int it = 0;; ++it) {
    ++ntile;
    const bool more = vb + nwg < ntiles;
    int ntb = 0, ny0 = 0, nx0 = 0;
    // ---- slot 1
    if (more) {                                        // the next tile's patch streams in under this tile
      step_tile(tb, ty, tx);                           // (tb / ty / tx run one tile ahead of y0 / x0 from here on)
      ntb = tb; ny0 = ty * TH; nx0 = tx * TW;
      if (team == 1) issue_patch(ntb, ny0, nx0, (it + 1) & 1, 0, G_SPLIT);
    }
    S2_STAMP(0)   // decode + DMA issue
    if (team == 0) {
      if (it > 0) stage2(ptb, py0, px0);
      S2_STAMP(1)   // S2 (team 0)
      k_loop(smem + (it & 1) * PATCH_BYTES);
      S2_STAMP(2)   // K (team 0)
    } else {
      if (it > 0) z_write();
      S2_STAMP(3)   // Z (team 1)
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    S2_STAMP(4)   // barrier 1
    // ---- slot 2
    if (team == 0) {
      if (more) issue_patch(ntb, ny0, nx0, (it + 1) & 1, G_SPLIT, NPIECES);   // (before Z: the pieces land under the SiLUs)
      S2_STAMP(0)   // DMA issue (team 0)
      z_write();
      S2_STAMP(3)   // Z (team 0)
    } else {
      if (it > 0) stage2(ptb, py0, px0);
      S2_STAMP(1)   // S2 (team 1)
      k_loop(smem + (it & 1) * PATCH_BYTES);
      S2_STAMP(2)   // K (team 1)
    }
    // the next patch has landed for this wave: team 1's pieces are older than its two stores of this slot; team 0's are the
    // youngest operations it has in flight
    if (team == 1 && it > 0) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    S2_STAMP(5)   // wait patch
    __builtin_amdgcn_s_barrier();
    S2_STAMP(6)   // barrier 2
    ptb = ctb; py0 = y0; px0 = x0;
    if (!more) break;
    vb += nwg;
    ctb = ntb; y0 = ny0; x0 = nx0;
  }
  // ---- the last tile drains: team 0 S2, team 1 Z; barrier; team 1 S2
  if (team == 0) stage2(ptb, py0, px0);
  else z_write();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (team == 1) stage2(ptb, py0, px0);
  if (stamps && lane == 0) {
    unsigned long long* o = stamps + ((long)blockIdx.x * NWAVES + wave) * 8;
    for (int k = 0; k < 7; ++k) o[k] = tacc[k];
    o[7] = (unsigned long long)ntile;
  }
#undef S2_STAMP
}

}  // namespace

// Eligibility: 3x3 / s2 / p1, 64 -> 128, SiLU, then 1x1 128 -> 128, SiLU; fragment-ordered weights; output map a multiple of
// the 8 x 8 tile and exactly half the input map; 31-bit byte offsets.
bool conv_s2c64_cv1_ok(const ConvArgs& a) {
  if (a.ksize != 3 || a.stride != 2 || a.pad != 1 || a.out_f32 || a.convt_co > 0 || a.tmode || a.phase || a.csplit || a.dec_preds || a.res)
    return false;
  if (a.Cin != 64 || a.Cout != 128 || a.cout2 != 128 || !a.wf || !a.wf2 || !a.bias || !a.bias2 || !a.act) return false;
  if (a.ldx % 8 || a.ldy % 8 || a.Hi != 2 * a.Ho || a.Wi != 2 * a.Wo || a.Ho % TH || a.Wo % TW) return false;
  const long nimg = a.Ho * a.Wo > 0 ? a.M / ((long)a.Ho * a.Wo) : 0;
  if (nimg < 1) return false;
  return ((nimg - 1) * a.x_bstride + (long)a.Hi * a.Wi * a.ldx) * 2 < (1L << 31);
}

int launch_conv_s2c64_cv1(const ConvArgs& a, hipStream_t s) {
  if (!conv_s2c64_cv1_ok(a)) return -1;
  const int tiles_x = a.Wo / TW, tiles_y = a.Ho / TH;
  const int B = a.M / (a.Ho * a.Wo);
  const int ntiles = B * tiles_y * tiles_x;
  static int slots = 0;
  if (!slots) {
    hipError_t e = hipFuncSetAttribute((const void*)conv3x3_s2c64_cv1_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
      return -2;
    slots = cus & ~7;
    if (slots < 8) slots = 8;
  }
  const int grid = ntiles <= slots ? ntiles : slots;
  const int step = grid >> 3;                          // tiles between two visits of a block (it walks only when grid = slots, % 8 == 0)
  const int sx = step % tiles_x, sy = (step / tiles_x) % tiles_y, sb = step / tiles_x / tiles_y;
  static const int prio = getenv("M355_S2C64_PRIO") ? atoi(getenv("M355_S2C64_PRIO")) : 0;   // experiment: s_setprio(1) around the K loop
  // diagnostic: M355_S2C64_STAMPS=<file> -> per-wave section cycles of the LAST launch, written after a stream sync [sync]
  static const char* st_path = getenv("M355_S2C64_STAMPS");
  static unsigned long long* d_st = nullptr;
  if (st_path && !d_st) {
    if (hipMalloc((void**)&d_st, (size_t)slots * NWAVES * 64) != hipSuccess) return -2;
    (void)hipMemset(d_st, 0, (size_t)slots * NWAVES * 64);
  }
  hipLaunchKernelGGL(conv3x3_s2c64_cv1_kernel, dim3(grid), dim3(64 * NWAVES), LDS_BYTES, s, a, tiles_x, tiles_y, ntiles, sx, sy, sb, prio, d_st);
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
