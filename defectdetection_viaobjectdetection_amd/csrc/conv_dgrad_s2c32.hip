// Input gradient of a 3x3 / stride-2 / pad-1 convolution with 32 input and 64 output channels (YOLOv8s model.1) on gfx950 MFMA.
//
// Replaces (SURVEY.md A13 backward): the dX half of aten conv2d's autograd for model.1, reached from SegmentationTrainer's backward
// (call site /root/reference/BscanBased/yolo_seg_train.py:12).  Its dX is the 320 x 320 x 32 map of a 640 x 640 batch -- 420 MB at
// batch 64 -- and it is the last convolution on the critical path of a training step (only layer 0's batch-norm backward and weight
// gradient follow).  On the im2col kernel (four phase convs with all four phases in one 128-channel tile, K = 256 per tile) it took
// 523-534 us: 12 800 tiles of 8 K steps each, prologue-bound.
//
// Here a WAVE owns a stream of chunks -- 32 consecutive dY pixels of one row i, i.e. 2 x 64 dX pixels (rows 2i, 2i + 1):
//   * dY rows i and i + 1 (33 pixels each: the right neighbour of the last one) go to a wave-private LDS image [2][33][64 ch]
//     through registers; the next chunk's loads are in flight while this one is multiplied;
//   * the 36 weight fragments (9 taps x 4 K slices of 16 output channels; A operand rows = the 32 dX channels in the row order that
//     leaves 16 consecutive channels to a lane-half) stay in registers for the whole stream -- they are read ONCE per wave from the
//     phase-form packed matrix m355_conv_launch(tmode = 2) already takes;
//   * phase (a, b) = dX pixels (2i + a, 2j + b): its (1 + a)(1 + b) taps x 4 slices are one accumulation chain of 32x32x16 MFMAs
//     (4 / 8 / 8 / 16: exactly the nine taps, no masked work), B fragments are plain 16-byte LDS reads (64 channels of a dY pixel
//     are contiguous), the result leaves as two 16-byte stores per lane.
// No accumulation into an existing gradient (the caller falls back), no atomics: bitwise reproducible.
#include <stdlib.h>

#include "common.h"

namespace m355 {
namespace {

typedef float float16v __attribute__((ext_vector_type(16)));
constexpr int DG_PX = 32;                   // dY pixels per chunk
constexpr int DG_COLS = DG_PX + 1;          // ... plus the right neighbour
constexpr int DG_WBYTES = 2 * DG_COLS * 128;   // one wave's image: [row 0/1][33 px][64 ch] fp16

__device__ __forceinline__ int row_plain(int rho) { return 16 * ((rho >> 2) & 1) + 4 * (rho >> 3) + (rho & 3); }

__global__ __launch_bounds__(256) void dgrad_s2c32_kernel(const half_t* dz, long dz_bs, int lddz, const half_t* wq, int kpad, half_t* dx,
                                                          long dx_bs, int lddx, int B, int Ho, int Wo) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, kg = lane >> 5;
  char* const zb = smem + wave * DG_WBYTES;
  const int segs = Wo / DG_PX;
  const long total = (long)B * Ho * segs, nwv = (long)gridDim.x * 4;

  // ---- weights: phase q = 2a + b, window slot ty * 2 + tx, K slice s -> rows q * 32 + ci, columns slot * 64 + 16 s + 8 kg ..
  // fragment index: 0..3 phase 0 (1 tap), 4..11 phase 1 (2), 12..19 phase 2 (2), 20..35 phase 3 (4)
  half8 A[36];
  {
    const int ci = row_plain(l31);
    int f = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int ty = 0; ty <= (q >> 1); ++ty)
#pragma unroll
        for (int tx = 0; tx <= (q & 1); ++tx)
#pragma unroll
          for (int s = 0; s < 4; ++s)
            A[f++] = *(const half8*)(wq + (long)(q * 32 + ci) * kpad + (ty * 2 + tx) * 64 + 16 * s + 8 * kg);
  }

  half8 zr[9];
  auto fetch = [&](long c) __attribute__((always_inline)) {
    const int seg = (int)(c % segs);
    const long r = c / segs;
    const int i = (int)(r % Ho);
    const long b = r / Ho;
    const int j0 = seg * DG_PX;
    const half_t* zim = dz + b * dz_bs;
#pragma unroll
    for (int u = 0; u < 9; ++u) {
      const int q = u * 64 + lane;                       // 16-byte chunk of the image: (row, pixel, 8-channel group)
      const int row = q / (DG_COLS * 8), rem = q - row * (DG_COLS * 8);
      const int px = rem >> 3, cg = rem & 7;
      zr[u] = half8{0, 0, 0, 0, 0, 0, 0, 0};
      if (row < 2 && i + row < Ho && j0 + px < Wo) zr[u] = *(const half8*)(zim + ((long)(i + row) * Wo + j0 + px) * lddz + cg * 8);
    }
  };
  // LDS image: pixel rows of 128 bytes; 16-byte chunk cg of pixel px sits at chunk position cg ^ (px & 7) -- the 16 lanes of a
  // ds_read_b128 service group read 16 consecutive pixels (128 bytes apart: two bank halves) at one chunk index: with the XOR they
  // cover all sixteen 16-byte bank slots
  auto stash = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < 9; ++u) {
      const int q = u * 64 + lane;
      const int pxr = q >> 3, cg = q & 7;                  // pxr = row * 33 + pixel
      const int px = pxr >= DG_COLS ? pxr - DG_COLS : pxr;
      if (q < 2 * DG_COLS * 8) *(half8*)(zb + pxr * 128 + ((cg ^ (px & 7)) << 4)) = zr[u];
    }
  };

  long c = (long)blockIdx.x * 4 + wave;
  if (c < total) fetch(c);
  for (; c < total; c += nwv) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (this wave's reads of the previous chunk's image are done)
    stash();
    const int seg = (int)(c % segs);
    const long r = c / segs;
    const int i = (int)(r % Ho);
    const long b = r / Ho;
    const int j0 = seg * DG_PX;
    if (c + nwv < total) fetch(c + nwv);
    half_t* const xrow = dx + b * dx_bs + ((long)(2 * i) * (2 * Wo) + 2 * (j0 + l31)) * lddx + 16 * kg;
    int f = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float16v acc;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
      for (int ty = 0; ty <= (q >> 1); ++ty)
#pragma unroll
        for (int tx = 0; tx <= (q & 1); ++tx)
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const half8 bf = *(const half8*)(zb + (ty * DG_COLS + l31 + tx) * 128 + (((2 * s + kg) ^ ((l31 + tx) & 7)) << 4));
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[f++], bf, acc, 0, 0, 0);
          }
      half8 o0, o1;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        o0[e] = m355_to_half(acc[e]);
        o1[e] = m355_to_half(acc[8 + e]);
      }
      half_t* const yp = xrow + ((long)(q >> 1) * (2 * Wo) + (q & 1)) * lddx;
      *(half8*)yp = o0;
      *(half8*)(yp + 8) = o1;
    }
  }
}

}  // namespace

bool dgrad_s2c32_ok(const ConvArgs& a) {
  static const bool off = getenv("M355_NO_DGRAD_S2C32") != nullptr;
  return !off && a.phase == 2 && a.ksize == 2 && a.convt_co == 32 && a.Cin == 64 && a.Cout == 128 && !a.res && !a.act && !a.out_f32 &&
         a.Kpad >= 256 && a.Kpad % 8 == 0 && a.ldx % 8 == 0 && a.ldy % 8 == 0 && a.Wo % DG_PX == 0 && a.Hi == a.Ho && a.Wi == a.Wo &&
         a.M % ((long)a.Ho * a.Wo) == 0;
}

int launch_dgrad_s2c32(const ConvArgs& a, hipStream_t s) {
  if (!dgrad_s2c32_ok(a)) return -1;
  const int B = (int)(a.M / ((long)a.Ho * a.Wo));
  const long chunks = (long)B * a.Ho * (a.Wo / DG_PX);
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return -2;
    if (cus < 1) cus = 1;
  }
  long grid = (long)cus * 2;                               // two blocks per CU: 248 registers per wave, 34 KB of LDS per block
  if (grid * 4 > chunks) grid = (chunks + 3) / 4;
  hipLaunchKernelGGL(dgrad_s2c32_kernel, dim3((unsigned)grid), dim3(256), 4 * DG_WBYTES, s, a.x, a.x_bstride, a.ldx, a.w, a.Kpad, (half_t*)a.y,
                     a.y_bstride, a.ldy, B, a.Ho, a.Wo);
  return (int)hipGetLastError();
}

}  // namespace m355
