// Train-mode building blocks: BatchNorm2d with batch statistics + SiLU, forward and backward,
// on NHWC fp16 activations with fp32 statistics.  Replaces aten batch_norm / silu (+ their autograd) reached
// from SegmentationModel.forward in training mode (call site BscanBased/yolo_seg_train.py:12).
// All HBM-bound: one 16-byte chunk (8 channels) per lane, per-channel fp32 partial sums in registers,
// block reduction through LDS, then a DETERMINISTIC cross-block reduction: every block stores its 2C partial sums
// to a workspace row and a small second kernel (one block per 64 columns) adds the rows in a fixed order.  No float
// atomics: two runs of one batch give the same bits.  (An in-kernel "last ticket adds the rows" form was measured at
// ~100 us per launch: one block reading 256 rows is a serial tail; the separate kernel costs what the memset of the
// atomic version did -- one more stream operation.)
#include "common.h"
#include "../../include/mi355yolo.h"

namespace m355 {
namespace {

constexpr int BN_THREADS = 256;

// v_exp + v_rcp (1 ulp), like the inference epilogues' m355_silu: the IEEE division of 1 / (1 + __expf(-v)) is ~10 VALU
// instructions per element and made the apply / reduce passes VALU-bound (92 us for 105 M elements = their HBM time)
__device__ __forceinline__ float sigmoid_f(float v) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.4426950408889634f)); }
constexpr int BN_ROWS = 512;   // most partial rows (= blocks) of a reduction launch

constexpr int WS_HEAD = 4;   // floats in front of the partial rows (kept 16-byte aligned)

// Tail of the two reduction kernels: this block's 2C partial sums (red[] holds every thread's 16) go to row blockIdx.x
// of the workspace with plain stores; bn_finalize_kernel adds the rows in block order.
__device__ __forceinline__ void store_block_partials(const float* red, int C, int cg, int lanes_px, float* ws) {
  float* const part = ws + WS_HEAD;
  const int n2 = 2 * C;
  for (int t = threadIdx.x; t < n2; t += BN_THREADS) {
    const int kind = t / C, c = t - kind * C;
    const int g = c / 8, j = c - g * 8;
    float acc = 0.f;
    for (int l = 0; l < lanes_px; ++l) acc += red[(l * cg + g) * 16 + kind * 8 + j];
    part[(long)blockIdx.x * n2 + t] = acc;
  }
}

// out[t] = sum over rows b = 0 .. rows-1 of part[b][t], FIXED association: wave w of 16 adds rows w, w+16, ... in order into
// one accumulator (eight loads in flight), then the sixteen wave sums are added pairwise ((0+1)+(2+3))+...  One block per
// 64 columns.  (Four waves per block walked 128 rows each: 7 us of pure latency, 132 times per training step.)
constexpr int FIN_WAVES = 16;
__global__ __launch_bounds__(FIN_WAVES * 64) void bn_finalize_kernel(const float* ws, int n2, int rows, float* out) {
  __shared__ float wsum[FIN_WAVES][64];
  const float* const part = ws + WS_HEAD;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + lane;
  float acc = 0.f;
  if (col < n2) {
    int b = w;
    for (; b + 7 * FIN_WAVES < rows; b += 8 * FIN_WAVES) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = part[(long)(b + FIN_WAVES * u) * n2 + col];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; b < rows; b += FIN_WAVES) acc += part[(long)b * n2 + col];
  }
  wsum[w][lane] = acc;
  __syncthreads();
#pragma unroll
  for (int h = FIN_WAVES / 2; h >= 1; h >>= 1) {
    if (w < h) wsum[w][lane] += wsum[w + h][lane];
    __syncthreads();
  }
  if (w == 0 && col < n2) out[col] = wsum[0][lane];
}

// partial rows of sum_px z (columns 0:C) and sum_px z^2 (C:2C)   (ws: WS_HEAD + gridDim.x * 2C floats)
__global__ __launch_bounds__(BN_THREADS) void bn_stats_kernel(const half_t* z, long npix, int ld, int C, float* ws) {
  extern __shared__ float red[];  // [BN_THREADS][16]
  const int cg = C / 8;                       // channel groups
  const int lanes_px = BN_THREADS / cg;       // pixel lanes per block (threads beyond lanes_px * cg idle)
  const int cgi = threadIdx.x % cg, pl = threadIdx.x / cg;
  float s[8], q[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) s[j] = q[j] = 0.f;
  if (pl < lanes_px) {
    // four pixels per trip: four independent 16-byte loads in flight per thread (one per trip ran at 1.4 TB/s)
    const long stride = (long)gridDim.x * lanes_px;
    const half_t* zp = z + cgi * 8;
    long p = (long)blockIdx.x * lanes_px + pl;
    for (; p + 3 * stride < npix; p += 4 * stride) {
      half8 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = *(const half8*)(zp + (p + u * stride) * ld);
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float f = (float)v[u][j];
          s[j] += f;
          q[j] += f * f;
        }
    }
    for (; p < npix; p += stride) {
      const half8 v = *(const half8*)(zp + p * ld);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float f = (float)v[j];
        s[j] += f;
        q[j] += f * f;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    red[threadIdx.x * 16 + j] = s[j];
    red[threadIdx.x * 16 + 8 + j] = q[j];
  }
  __syncthreads();
  store_block_partials(red, C, cg, lanes_px, ws);
}

// mean / invstd from the sums (biased variance, like torch batch_norm in training mode), optional
// running-stat update with momentum (unbiased variance), then y = silu(gamma * (z - mean) * invstd + beta).
// A thread owns one 8-channel group for its whole life (scale / shift live in registers) and walks pixels.
__global__ __launch_bounds__(BN_THREADS) void bn_silu_apply_kernel(const half_t* z, long npix, int ldz, int C,
                                                                   const float* sums, const float* gamma,
                                                                   const float* beta, float eps, half_t* y, int ldy,
                                                                   const half_t* res, int ldr, float* mean_out,
                                                                   float* invstd_out, int act, float* run_mean,
                                                                   float* run_var, float momentum) {
  const int cg = C / 8;
  const int lanes_px = BN_THREADS / cg;
  const int cgi = threadIdx.x % cg, pl = threadIdx.x / cg;
  if (pl >= lanes_px) return;
  const float inv_n = 1.0f / (float)npix;
  float sc[8], sh[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = cgi * 8 + j;
    const float m = sums[c] * inv_n;
    const float var = fmaxf(sums[C + c] * inv_n - m * m, 0.f);
    const float is = rsqrtf(var + eps);
    sc[j] = gamma[c] * is;
    sh[j] = beta[c] - m * sc[j];
    if (blockIdx.x == 0 && pl == 0) {
      if (mean_out) {
        mean_out[c] = m;
        invstd_out[c] = is;
      }
      if (run_mean) {  // running statistics: momentum update with the unbiased variance
        const float n = (float)npix;
        run_mean[c] = (1.0f - momentum) * run_mean[c] + momentum * m;
        run_var[c] = (1.0f - momentum) * run_var[c] + momentum * var * (n / fmaxf(n - 1.0f, 1.0f));
      }
    }
  }
  const long stride = (long)gridDim.x * lanes_px;
  constexpr int U = 4;
  for (long p0 = (long)blockIdx.x * lanes_px + pl; p0 < npix; p0 += stride * U) {
    half8 v[U], rv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long p = p0 + u * stride;
      if (p < npix) {
        v[u] = *(const half8*)(z + p * ldz + cgi * 8);
        if (res) rv[u] = *(const half8*)(res + p * ldr + cgi * 8);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long p = p0 + u * stride;
      if (p < npix) {
        half8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float t = (float)v[u][j] * sc[j] + sh[j];
          float r = act ? t * sigmoid_f(t) : t;
          if (res) r += (float)rv[u][j];
          o[j] = (half_t)r;
        }
        *(half8*)(y + p * ldy + cgi * 8) = o;
      }
    }
  }
}

// backward reductions: rsum[0:C] = sum du, rsum[C:2C] = sum du * xhat, du = dy * silu'(u)
__global__ __launch_bounds__(BN_THREADS) void bn_silu_bwd_reduce_kernel(const half_t* z, const half_t* dy, long npix,
                                                                        int ldz, int lddy, int C, const float* mean,
                                                                        const float* invstd, const float* gamma,
                                                                        const float* beta, int act, float* ws) {
  extern __shared__ float red[];
  const int cg = C / 8;
  const int lanes_px = BN_THREADS / cg;
  const int cgi = threadIdx.x % cg, pl = threadIdx.x / cg;
  float s[8], q[8], m[8], is[8], ga[8], be[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    s[j] = q[j] = 0.f;
    const int c = cgi * 8 + j;
    m[j] = mean[c]; is[j] = invstd[c]; ga[j] = gamma[c]; be[j] = beta[c];
  }
  if (pl < lanes_px) {
    auto accumulate = [&](const half8& v, const half8& d) __attribute__((always_inline)) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float xh = ((float)v[j] - m[j]) * is[j];
        float du = (float)d[j];
        if (act) {
          const float u = ga[j] * xh + be[j];
          const float sg = sigmoid_f(u);
          du *= sg * (1.0f + u * (1.0f - sg));
        }
        s[j] += du;
        q[j] += du * xh;
      }
    };
    // four pixels per trip: eight independent 16-byte loads in flight per thread
    const long stride = (long)gridDim.x * lanes_px;
    const half_t* zp = z + cgi * 8;
    const half_t* dp = dy + cgi * 8;
    long p = (long)blockIdx.x * lanes_px + pl;
    for (; p + 3 * stride < npix; p += 4 * stride) {
      half8 v[4], d[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        v[u] = *(const half8*)(zp + (p + u * stride) * ldz);
        d[u] = *(const half8*)(dp + (p + u * stride) * lddy);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) accumulate(v[u], d[u]);
    }
    for (; p < npix; p += stride) accumulate(*(const half8*)(zp + p * ldz), *(const half8*)(dp + p * lddy));
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    red[threadIdx.x * 16 + j] = s[j];
    red[threadIdx.x * 16 + 8 + j] = q[j];
  }
  __syncthreads();
  store_block_partials(red, C, cg, lanes_px, ws);
}

// dz = gamma * invstd * (du - dbeta / N - xhat * dgamma / N); same thread-owns-a-channel-group walk as the forward
__global__ __launch_bounds__(BN_THREADS) void bn_silu_bwd_apply_kernel(const half_t* z, const half_t* dy, long npix,
                                                                       int ldz, int lddy, int C, const float* mean,
                                                                       const float* invstd, const float* gamma,
                                                                       const float* beta, const float* rsum,
                                                                       half_t* dz, int lddz, int act) {
  const int cg = C / 8;
  const int lanes_px = BN_THREADS / cg;
  const int cgi = threadIdx.x % cg, pl = threadIdx.x / cg;
  if (pl >= lanes_px) return;
  const float inv_n = 1.0f / (float)npix;
  float m[8], is[8], ga[8], be[8], k1[8], k2[8], k3[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = cgi * 8 + j;
    m[j] = mean[c]; is[j] = invstd[c]; ga[j] = gamma[c]; be[j] = beta[c];
    k1[j] = ga[j] * is[j];
    k2[j] = rsum[c] * inv_n;
    k3[j] = rsum[C + c] * inv_n;
  }
  const long stride = (long)gridDim.x * lanes_px;
  constexpr int U = 4;
  for (long p0 = (long)blockIdx.x * lanes_px + pl; p0 < npix; p0 += stride * U) {
    half8 v[U], d[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long p = p0 + u * stride;
      if (p < npix) {
        v[u] = *(const half8*)(z + p * ldz + cgi * 8);
        d[u] = *(const half8*)(dy + p * lddy + cgi * 8);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long p = p0 + u * stride;
      if (p < npix) {
        half8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float xh = ((float)v[u][j] - m[j]) * is[j];
          float du = (float)d[u][j];
          if (act) {
            const float t = ga[j] * xh + be[j];
            const float sg = sigmoid_f(t);
            du *= sg * (1.0f + t * (1.0f - sg));
          }
          o[j] = (half_t)(k1[j] * (du - k2[j] - xh * k3[j]));
        }
        *(half8*)(dz + p * lddz + cgi * 8) = o;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// Optimizer step over the flat fp32 parameter buffer: one pass reads p, g, state and the EMA
// copy and writes them back -- 36 B/param for AdamW, 28 B/param for SGD -- instead of ~10 elementwise passes.
// group[i]: 0 = conv weights (decayed), 1 = norm weights, 2 = biases (own learning rate during warm-up).
// grad_mul folds 1/loss_scale and the gradient-clipping coefficient.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adamw_step_kernel(float* p, const float* g, float* m, float* v, float* ema,
                                                         const unsigned char* group, long n, float lr, float lr_bias,
                                                         float beta1, float beta2, float eps, float wd, float bc1,
                                                         float rsqrt_bc2, float grad_mul, float ema_d) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int gr = group[i];
    const float l = gr == 2 ? lr_bias : lr;
    const float gi = g[i] * grad_mul;
    float pi = p[i];
    if (gr == 0) pi *= 1.0f - l * wd;
    const float mi = beta1 * m[i] + (1.0f - beta1) * gi;
    const float vi = beta2 * v[i] + (1.0f - beta2) * gi * gi;
    pi -= (l / bc1) * mi / (sqrtf(vi) * rsqrt_bc2 + eps);
    p[i] = pi; m[i] = mi; v[i] = vi;
    if (ema) ema[i] = ema_d * ema[i] + (1.0f - ema_d) * pi;
  }
}

__global__ __launch_bounds__(256) void sgd_step_kernel(float* p, const float* g, float* buf, float* ema,
                                                       const unsigned char* group, long n, float lr, float lr_bias,
                                                       float momentum, int nesterov, float wd, float grad_mul, float ema_d) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int gr = group[i];
    const float l = gr == 2 ? lr_bias : lr;
    float pi = p[i];
    float gi = g[i] * grad_mul;
    if (gr == 0) gi += wd * pi;
    const float bi = momentum * buf[i] + gi;
    gi = nesterov ? gi + momentum * bi : bi;
    pi -= l * gi;
    p[i] = pi; buf[i] = bi;
    if (ema) ema[i] = ema_d * ema[i] + (1.0f - ema_d) * pi;
  }
}

// out[0] = sum g^2 over the finite entries, out[1] = number of non-finite entries.  Deterministic: wave shuffle reduction,
// the four wave sums of a block added in wave order, one partial pair per block in out[8 + 2 b]; grad_sumsq_final_kernel
// (one block) adds them in block order.  out: SUMSQ_WS floats.
constexpr int SUMSQ_BLOCKS = 1024, SUMSQ_WS = 8 + 2 * SUMSQ_BLOCKS;
__global__ __launch_bounds__(256) void grad_sumsq_kernel(const float* g, long n, float* out) {
  __shared__ float wsum[8];
  float s = 0.f, bad = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float x = g[i];
    if (isfinite(x)) s += x * x; else bad += 1.f;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    s += __shfl_down(s, o, 64);
    bad += __shfl_down(bad, o, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    wsum[(threadIdx.x >> 6) * 2] = s;
    wsum[(threadIdx.x >> 6) * 2 + 1] = bad;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    out[8 + 2 * blockIdx.x] = ((wsum[0] + wsum[2]) + wsum[4]) + wsum[6];
    out[9 + 2 * blockIdx.x] = ((wsum[1] + wsum[3]) + wsum[5]) + wsum[7];
  }
}
__global__ __launch_bounds__(256) void grad_sumsq_final_kernel(float* out, int nblocks) {
  __shared__ float ts[256], tb[256];
  float a0 = 0.f, a1 = 0.f;
  for (int b = threadIdx.x * 4; b < threadIdx.x * 4 + 4 && b < nblocks; ++b) {   // 4 consecutive blocks per thread, in order
    a0 += out[8 + 2 * b];
    a1 += out[9 + 2 * b];
  }
  ts[threadIdx.x] = a0;
  tb[threadIdx.x] = a1;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t0 = 0.f, t1 = 0.f;
    for (int i = 0; i < 256; ++i) { t0 += ts[i]; t1 += tb[i]; }               // thread partials in thread order
    out[0] = t0;
    out[1] = t1;
  }
}

int grid_for(long work_items) {
  long b = (work_items + BN_THREADS - 1) / BN_THREADS;
  if (b > 256 * 8) b = 256 * 8;
  if (b < 1) b = 1;
  return (int)b;
}

// blocks for the pixel-walking apply kernels: 4 pixels per thread per trip, at most 16 blocks per CU
// reduction kernels: every block writes one partial row; two blocks per CU (one left HBM at 3.1 - 3.7 TB/s)
int grid_red(long work_items) {
  const int g = grid_for(work_items);
  return g > BN_ROWS ? BN_ROWS : g;
}

int grid_px(long npix, int lanes_px) {
  // at least 16 pixels per thread: a thread's prologue is ~30 dependent-latency loads of per-channel constants, and with 4
  // pixels each (C = 256: 8 pixel lanes per block) the apply pass ran at 1.8 TB/s
  long b = (npix + (long)lanes_px * 16 - 1) / ((long)lanes_px * 16);
  if (b > 256 * 8) b = 256 * 8;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace

size_t bn_workspace_floats(int C) { return (size_t)WS_HEAD + (size_t)BN_ROWS * 2 * C + 2 * C; }

// ws: bn_workspace_floats(C) floats (per-block partial rows, then the 2C batch sums at ws[WS_HEAD + BN_ROWS * 2C ...])
int launch_bn_silu_train_fwd(const half_t* z, long npix, int ldz, int C, const float* gamma, const float* beta,
                             float eps, half_t* y, int ldy, const half_t* res, int ldr, float* sums, float* mean_out,
                             float* invstd_out, int act, float* run_mean, float* run_var, float momentum, hipStream_t s) {
  if (C % 8 || ldz % 8 || ldy % 8 || C / 8 > BN_THREADS) return -1;
  const int lanes_px = BN_THREADS / (C / 8);
  float* const ws = sums;
  float* const tot = ws + WS_HEAD + (size_t)BN_ROWS * 2 * C;
  const int gr = grid_red(npix * BN_THREADS / lanes_px / 4);
  hipLaunchKernelGGL(bn_stats_kernel, dim3(gr), dim3(BN_THREADS), BN_THREADS * 16 * sizeof(float), s, z, npix, ldz, C, ws);
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((2 * C + 63) / 64), dim3(FIN_WAVES * 64), 0, s, ws, 2 * C, gr, tot);
  hipLaunchKernelGGL(bn_silu_apply_kernel, dim3(grid_px(npix, lanes_px)), dim3(BN_THREADS), 0, s, z, npix, ldz, C, tot,
                     gamma, beta, eps, y, ldy, res, ldr, mean_out, invstd_out, act, run_mean, run_var, momentum);
  return (int)hipGetLastError();
}

// rsum: device float[2C]; on return rsum[0:C] = dbeta, rsum[C:2C] = dgamma.  ws: as for the forward launch.
int launch_bn_silu_train_bwd(const half_t* z, const half_t* dy, long npix, int ldz, int lddy, int C, const float* mean,
                             const float* invstd, const float* gamma, const float* beta, float* rsum, half_t* dz,
                             int lddz, int act, float* ws, hipStream_t s) {
  if (C % 8 || ldz % 8 || lddy % 8 || lddz % 8 || C / 8 > BN_THREADS || !ws) return -1;
  const int lanes_px = BN_THREADS / (C / 8);
  const int gr = grid_red(npix * BN_THREADS / lanes_px / 4);
  hipLaunchKernelGGL(bn_silu_bwd_reduce_kernel, dim3(gr), dim3(BN_THREADS), BN_THREADS * 16 * sizeof(float), s, z, dy, npix, ldz,
                     lddy, C, mean, invstd, gamma, beta, act, ws);
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((2 * C + 63) / 64), dim3(FIN_WAVES * 64), 0, s, ws, 2 * C, gr, rsum);
  hipLaunchKernelGGL(bn_silu_bwd_apply_kernel, dim3(grid_px(npix, lanes_px)), dim3(BN_THREADS), 0, s, z, dy, npix, ldz,
                     lddy, C, mean, invstd, gamma, beta, rsum, dz, lddz, act);
  return (int)hipGetLastError();
}

int launch_adamw_step(float* p, const float* g, float* m, float* v, float* ema, const unsigned char* group, long n,
                      float lr, float lr_bias, float beta1, float beta2, float eps, float wd, int step, float grad_mul,
                      float ema_d, hipStream_t s) {
  if (n <= 0 || step < 1) return -1;
  const float bc1 = 1.0f - powf(beta1, (float)step);
  const float rsqrt_bc2 = 1.0f / sqrtf(1.0f - powf(beta2, (float)step));
  hipLaunchKernelGGL(adamw_step_kernel, dim3(grid_for(n)), dim3(256), 0, s, p, g, m, v, ema, group, n, lr, lr_bias, beta1,
                     beta2, eps, wd, bc1, rsqrt_bc2, grad_mul, ema_d);
  return (int)hipGetLastError();
}

int launch_sgd_step(float* p, const float* g, float* buf, float* ema, const unsigned char* group, long n, float lr,
                    float lr_bias, float momentum, int nesterov, float wd, float grad_mul, float ema_d, hipStream_t s) {
  if (n <= 0) return -1;
  hipLaunchKernelGGL(sgd_step_kernel, dim3(grid_for(n)), dim3(256), 0, s, p, g, buf, ema, group, n, lr, lr_bias, momentum,
                     nesterov, wd, grad_mul, ema_d);
  return (int)hipGetLastError();
}

size_t grad_sumsq_workspace_floats() { return SUMSQ_WS; }

int launch_grad_sumsq(const float* g, long n, float* out, hipStream_t s) {
  if (n <= 0) return -1;
  int blocks = grid_for(n);
  if (blocks > SUMSQ_BLOCKS) blocks = SUMSQ_BLOCKS;
  hipLaunchKernelGGL(grad_sumsq_kernel, dim3(blocks), dim3(256), 0, s, g, n, out);
  hipLaunchKernelGGL(grad_sumsq_final_kernel, dim3(1), dim3(256), 0, s, out, blocks);
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// Backward of SPPF's pooling chain y1 = mp5(a), y2 = mp5(y1), y3 = mp5(y2) (MaxPool2d(5, 1, 2); SURVEY A13, upstream reaches
// it through autograd of F.max_pool2d).  The training step ran it as three torch max_pool2d forward + backward pairs on
// fp32 NCHW copies (1.2 ms of a 44 ms step).  One block per (image, 8-channel group): the whole map of those channels lives
// in LDS.  Per stage: the argmax of every window (first maximum in row-major order, torch's rule), then a GATHER -- position
// p sums the incoming gradient of the windows whose argmax is p, in ascending (row, column) order of the windows -- so
// there are no atomics and the result is bitwise reproducible.  gy (B,H,W,>=3C): gradients of [y1 | y2 | y3]; ga receives
// d(loss)/da, stored or accumulated (fp16).
// ---------------------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void sppf_pool_bwd_kernel(const half_t* a, long a_bs, int lda, const half_t* y, long y_bs, int ldy,
                                                            const half_t* gy, long gy_bs, int ldgy, half_t* ga, long ga_bs,
                                                            int ldga, int H, int W, int C, int accumulate) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int n = H * W * 8;
  half_t* v = (half_t*)smem;                       // [px][8] values of the stage's input map
  unsigned short* idx = (unsigned short*)(v + n);  // [window][8] argmax pixel of every window
  float* rout = (float*)(idx + n);                 // [px][8] gradient w.r.t. the stage's OUTPUT map
  float* rin = rout + n;                           // [px][8] gradient w.r.t. the stage's INPUT map
  const int groups = C / 8;
  const int b = blockIdx.x / groups, c0 = (blockIdx.x - b * groups) * 8;
  const half_t* ab = a + (long)b * a_bs + c0;
  const half_t* yb = y + (long)b * y_bs + c0;
  const half_t* gb = gy + (long)b * gy_bs + c0;
  for (int i = threadIdx.x; i < H * W; i += 256) {           // gradient of y3
    const half8 g3 = *(const half8*)(gb + (long)i * ldgy + 2 * C);
#pragma unroll
    for (int j = 0; j < 8; ++j) rout[i * 8 + j] = (float)g3[j];
  }
  // work item = (pixel, four of the eight channels): every LDS access of the two window walks is 8 or 16 bytes wide.  (One channel per
  // item, 2-byte reads: 150 scalar LDS reads per element and stage made this 26 MB operation 460 us of a 25 ms step.)
  typedef unsigned short ushort4v __attribute__((ext_vector_type(4)));
  const int items = H * W * 2;
  for (int stage = 2; stage >= 0; --stage) {
    const half_t* src = stage == 2 ? yb + C : (stage == 1 ? yb : ab);       // y2, y1, a
    const int lds_ = stage == 0 ? lda : ldy;
    for (int i = threadIdx.x; i < H * W; i += 256) *(half8*)(v + i * 8) = *(const half8*)(src + (long)i * lds_);
    __syncthreads();
    for (int e = threadIdx.x; e < items; e += 256) {         // argmax of the window centred on pixel q (first maximum, row-major)
      const int q = e >> 1, h4 = (e & 1) * 4;
      const int qy = q / W, qx = q - qy * W;
      float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
      const int first = (qy - 2 < 0 ? 0 : qy - 2) * W + (qx - 2 < 0 ? 0 : qx - 2);
      int bi[4] = {first, first, first, first};
#pragma unroll
      for (int dy = -2; dy <= 2; ++dy) {
        const int yy = qy + dy;
        if (yy < 0 || yy >= H) continue;
#pragma unroll
        for (int dx = -2; dx <= 2; ++dx) {
          const int xx = qx + dx;
          if (xx < 0 || xx >= W) continue;
          const half4 t = *(const half4*)(v + (yy * W + xx) * 8 + h4);
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if ((float)t[j] > best[j]) { best[j] = (float)t[j]; bi[j] = yy * W + xx; }
        }
      }
      *(ushort4v*)(idx + q * 8 + h4) = ushort4v{(unsigned short)bi[0], (unsigned short)bi[1], (unsigned short)bi[2], (unsigned short)bi[3]};
    }
    __syncthreads();
    for (int e = threadIdx.x; e < items; e += 256) {         // gather, windows in ascending (row, column) order
      const int p = e >> 1, h4 = (e & 1) * 4;
      const int py = p / W, px = p - py * W;
      float sum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int dy = -2; dy <= 2; ++dy) {
        const int yy = py + dy;
        if (yy < 0 || yy >= H) continue;
#pragma unroll
        for (int dx = -2; dx <= 2; ++dx) {
          const int xx = px + dx;
          if (xx < 0 || xx >= W) continue;
          const int q = yy * W + xx;
          const ushort4v id = *(const ushort4v*)(idx + q * 8 + h4);
          const float4v r = *(const float4v*)(rout + q * 8 + h4);
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (id[j] == (unsigned short)p) sum[j] += r[j];
        }
      }
      if (stage > 0) {   // the map's own slot in the concat: y2 (stage 2), y1 (stage 1)
        const half4 g = *(const half4*)(gb + (long)p * ldgy + (stage - 1) * C + h4);
#pragma unroll
        for (int j = 0; j < 4; ++j) sum[j] += (float)g[j];
      }
      *(float4v*)(rin + p * 8 + h4) = float4v{sum[0], sum[1], sum[2], sum[3]};
    }
    __syncthreads();
    float* t = rout; rout = rin; rin = t;
  }
  half_t* gab = ga + (long)b * ga_bs + c0;
  for (int i = threadIdx.x; i < H * W; i += 256) {
    half8 o;
    if (accumulate) {
      const half8 old = *(const half8*)(gab + (long)i * ldga);
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = m355_to_half((float)old[j] + (float)m355_to_half(rout[i * 8 + j]));
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = m355_to_half(rout[i * 8 + j]);
    }
    *(half8*)(gab + (long)i * ldga) = o;
  }
}
}  // namespace

int launch_sppf_pool_bwd(const half_t* a, long a_bs, int lda, const half_t* y, long y_bs, int ldy, const half_t* gy, long gy_bs,
                         int ldgy, half_t* ga, long ga_bs, int ldga, int B, int H, int W, int C, int accumulate, hipStream_t s) {
  if (C % 8 || lda % 8 || ldy % 8 || ldgy % 8 || ldga % 8 || H * W > 65535) return -1;
  const size_t lds = (size_t)H * W * 8 * (2 + 2 + 4 + 4);
  if (lds > 160 * 1024) return -1;
  if (lds > 65536) {
    hipError_t e = hipFuncSetAttribute((const void*)sppf_pool_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(sppf_pool_bwd_kernel, dim3(B * (C / 8)), dim3(256), lds, s, a, a_bs, lda, y, y_bs, ldy, gy, gy_bs, ldgy, ga,
                     ga_bs, ldga, H, W, C, accumulate);
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// Re-pack of the fp32 master weights into the fp16 GEMM layouts after an optimizer step (forward [Cout][(kh,kw,ci)], dgrad
// [Cin][(kh',kw',co)] with flipped taps, ConvTranspose forms): ~150 strided convert-copies per step were ~150 launch-bound
// torch kernels (1.8 ms of a 37 ms step).  Here every copy is a JOB -- a 4-d iteration space with signed element strides
// on both sides (a flip is a negative source stride) -- and one launch walks all jobs: block b works on 1024 consecutive
// elements of job block_job[b] (innermost dimension fastest = contiguous fp16 stores).
// ---------------------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void repack_kernel(const ::m355_repack_job* jobs, const int* block_job) {
  const ::m355_repack_job j = jobs[block_job[blockIdx.x]];
  const long total = (long)j.n[0] * j.n[1] * j.n[2] * j.n[3];
  const long base = (long)(blockIdx.x - j.block0) * 1024;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const long e = base + u * 256 + threadIdx.x;
    if (e >= total) break;
    long r = e;
    const int i3 = (int)(r % j.n[3]); r /= j.n[3];
    const int i2 = (int)(r % j.n[2]); r /= j.n[2];
    const int i1 = (int)(r % j.n[1]); r /= j.n[1];
    const int i0 = (int)r;
    const float v = ((const float*)j.src)[i0 * j.ss[0] + i1 * j.ss[1] + i2 * j.ss[2] + i3 * j.ss[3]];
    ((half_t*)j.dst)[i0 * j.ds[0] + i1 * j.ds[1] + i2 * j.ds[2] + i3 * j.ds[3]] = (half_t)v;
  }
}
}  // namespace

int launch_repack(const void* d_jobs, const int* d_block_job, int nblocks, hipStream_t s) {
  if (!d_jobs || !d_block_job || nblocks < 1) return -1;
  hipLaunchKernelGGL(repack_kernel, dim3(nblocks), dim3(256), 0, s, (const ::m355_repack_job*)d_jobs, d_block_job);
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// Gradient glue of the training step.  Each of these stood as 2-4 strided torch kernels with an fp32 copy of the
// whole tensor in between (rocprofv3 sequence of one step, tools/trace_seq.py): the bias gradients of the head's output
// convs and of the ConvTranspose as `x.float().sum(dims)` (0.7 + 0.5 ms for 0.3 GB of reads), the backward of the nearest
// 2x upsample as `.float().sum((2, 4))` + add (0.4 ms), the uint8 -> fp16 input conversion as four passes (0.55 ms).
// ---------------------------------------------------------------------------------------------------------
namespace {
constexpr int CS_BLOCKS = 2048;   // target number of partial rows of a column sum

// Partial column sums of rows [blockIdx.x * chunk, +chunk) of batch blockIdx.y of a (nb, rows, cols) fp32 view: thread t owns
// column t % cols on row lane t / cols (256 / cols lanes), adds its rows in order; the lanes are added in lane order.
__global__ __launch_bounds__(256) void colsum_f32_kernel(const float* src, long bstride, long rows, int ld, int cols, long chunk,
                                                         float* ws) {
  __shared__ float red[256];
  const int rl = 256 / cols;
  const int c = threadIdx.x % cols, l = threadIdx.x / cols;
  const long r0 = blockIdx.x * chunk, r1 = r0 + chunk < rows ? r0 + chunk : rows;
  const float* const p = src + blockIdx.y * bstride + c;
  float acc = 0.f;
  if (l < rl) {
    long r = r0 + l;
    for (; r + 3 * rl < r1; r += 4 * rl) {
      const float v0 = p[r * ld], v1 = p[(r + rl) * ld], v2 = p[(r + 2 * rl) * ld], v3 = p[(r + 3 * rl) * ld];
      acc += v0; acc += v1; acc += v2; acc += v3;
    }
    for (; r < r1; r += rl) acc += p[r * ld];
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  if (threadIdx.x < cols) {
    float t = 0.f;
    for (int k = 0; k < rl; ++k) t += red[k * cols + threadIdx.x];
    ws[WS_HEAD + ((long)blockIdx.y * gridDim.x + blockIdx.x) * cols + threadIdx.x] = t;
  }
}

// The same over an fp16 view with cols % 8 == 0: a thread owns an 8-channel group (16-byte loads), fp32 accumulators.
__global__ __launch_bounds__(256) void colsum_f16_kernel(const half_t* src, long bstride, long rows, int ld, int cols, long chunk,
                                                         float* ws) {
  __shared__ float red[256 * 8];
  const int cg = cols / 8, rl = 256 / cg;
  const int g = threadIdx.x % cg, l = threadIdx.x / cg;
  const long r0 = blockIdx.x * chunk, r1 = r0 + chunk < rows ? r0 + chunk : rows;
  const half_t* const p = src + blockIdx.y * bstride + g * 8;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (l < rl) {
    long r = r0 + l;
    for (; r + 7 * rl < r1; r += 8 * rl) {
      half8 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *(const half8*)(p + (r + (long)u * rl) * ld);
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += (float)v[u][j];
    }
    for (; r < r1; r += rl) {
      const half8 v = *(const half8*)(p + r * ld);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += (float)v[j];
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) red[threadIdx.x * 8 + j] = acc[j];
  __syncthreads();
  for (int t = threadIdx.x; t < cols; t += 256) {
    const int gg = t >> 3, j = t & 7;
    float sum = 0.f;
    for (int k = 0; k < rl; ++k) sum += red[(k * cg + gg) * 8 + j];
    ws[WS_HEAD + ((long)blockIdx.y * gridDim.x + blockIdx.x) * cols + t] = sum;
  }
}

// d[b][y][x][c] (=|+=) g[b][2y][2x][c] + g[b][2y][2x+1][c] + g[b][2y+1][2x][c] + g[b][2y+1][2x+1][c]: the four taps are added in
// fp32 in that order and rounded to fp16 once; accumulate adds that fp16 value to what d holds (fp32 add, one rounding) -- the
// arithmetic of `gs = g.float().sum(...)`, `d.copy_(gs)` / `d.add_(gs.half())`.
__global__ __launch_bounds__(256) void upsample2x_bwd_kernel(const half_t* g, long g_bs, int ldg, half_t* d, long d_bs, int ldd, int H,
                                                             int W, int C, int accumulate, long total) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const int cg = C / 8;
  const int c8 = (int)(e % cg);
  const long px = e / cg;
  const int x = (int)(px % W);
  const long by = px / W;
  const int y = (int)(by % H);
  const long b = by / H;
  const half_t* gp = g + b * g_bs + ((long)(2 * y) * (2 * W) + 2 * x) * ldg + c8 * 8;
  const half8 v00 = *(const half8*)gp, v01 = *(const half8*)(gp + ldg);
  const half8 v10 = *(const half8*)(gp + (long)2 * W * ldg), v11 = *(const half8*)(gp + (long)2 * W * ldg + ldg);
  half_t* dp = d + b * d_bs + ((long)y * W + x) * ldd + c8 * 8;
  half8 o;
  if (accumulate) {
    const half8 old = *(const half8*)dp;
#pragma unroll
    for (int j = 0; j < 8; ++j)
      o[j] = m355_to_half((float)old[j] + (float)m355_to_half((((float)v00[j] + (float)v01[j]) + (float)v10[j]) + (float)v11[j]));
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = m355_to_half((((float)v00[j] + (float)v01[j]) + (float)v10[j]) + (float)v11[j]);
  }
  *(half8*)dp = o;
}

// (N, 3) uint8 pixels -> (N, 8) fp16 rows [r/255, g/255, b/255, 0, 0, 0, 0, 0] (correctly rounded fp32 division, then one
// rounding to fp16 = `(u8.float() / 255).half()`).  A block converts 1024 pixels: their 3072 bytes go through LDS as coalesced
// dwords, then thread t writes pixels t, t + 256, ... so that a wave's store is 1 KB of consecutive rows.
__global__ __launch_bounds__(256) void u8_to_f16x8_kernel(const unsigned char* src, half_t* dst, long npx, int aligned) {
  __shared__ unsigned int stage[768];
  const long p0 = (long)blockIdx.x * 1024;
  const int np = npx - p0 >= 1024 ? 1024 : (int)(npx - p0);
  const unsigned char* const s = src + p0 * 3;
  unsigned char* const sb = (unsigned char*)stage;
  if (aligned && np == 1024) {
#pragma unroll
    for (int k = 0; k < 3; ++k) stage[k * 256 + threadIdx.x] = ((const unsigned int*)s)[k * 256 + threadIdx.x];
  } else {
    for (int i = threadIdx.x; i < np * 3; i += 256) sb[i] = s[i];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int p = k * 256 + threadIdx.x;
    if (p >= np) break;
    half8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (half_t)0.f;
#pragma unroll
    for (int j = 0; j < 3; ++j) o[j] = m355_to_half(__fdiv_rn((float)sb[p * 3 + j], 255.0f));
    *(half8*)(dst + (p0 + p) * 8) = o;
  }
}
}  // namespace

// ---------------------------------------------------------------------------------------------------------
// YOLOv9c training glue (the graph /root/reference/BscanBased/yolo_seg_train.py:7 names).  Two ops of that graph
// have no convolution in them and ran as torch expressions on fp32 NCHW copies until round 4:
//   RepConvN's tail        y = SiLU(a + b) of the two activation-free Conv + BN branches (upstream RepConvN.forward: act(conv1(x) + conv2(x)))
//   ADown's pooling front  t = avg_pool2d(x, 2, 1, 0);  p1 = t[:, :c];  p2 = max_pool2d(t[:, c:], 3, 2, 1)   (upstream ADown.forward)
// Forward and backward of each as one pass over NHWC fp16 rows, a thread per (pixel, 8-channel group).  The arithmetic keeps the
// rounding points of the torch form: v = fp16(a + b) is stored for the backward; averages are fp32 sums x 0.25 rounded to fp16 once;
// the max-pool's argmax (first maximum in row-major window order, torch's rule) is stored as one byte per element by the forward
// pass, so the backward is a GATHER in a fixed order: no atomics, bitwise reproducible.
// ---------------------------------------------------------------------------------------------------------
namespace {

__global__ __launch_bounds__(256) void addsilu_fwd_kernel(const half_t* a, const half_t* b, half_t* v, half_t* y, long npix, int ldy, int C,
                                                          long total) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const int cg = C / 8;
  const long p = e / cg;
  const int c8 = (int)(e - p * cg) * 8;
  const half8 va = *(const half8*)(a + p * C + c8), vb = *(const half8*)(b + p * C + c8);
  half8 s, o;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    s[j] = m355_to_half((float)va[j] + (float)vb[j]);
    const float f = (float)s[j];
    o[j] = m355_to_half(f * sigmoid_f(f));
  }
  *(half8*)(v + p * C + c8) = s;
  *(half8*)(y + p * ldy + c8) = o;
}

// g = dy * SiLU'(v): ONE buffer, the gradient of both branches
__global__ __launch_bounds__(256) void addsilu_bwd_kernel(const half_t* v, const half_t* dy, int lddy, half_t* g, int C, long total) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const int cg = C / 8;
  const long p = e / cg;
  const int c8 = (int)(e - p * cg) * 8;
  const half8 vv = *(const half8*)(v + p * C + c8), d = *(const half8*)(dy + p * lddy + c8);
  half8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float f = (float)vv[j], sg = sigmoid_f(f);
    o[j] = m355_to_half((float)d[j] * (sg * (1.0f + f * (1.0f - sg))));
  }
  *(half8*)(g + p * C + c8) = o;
}

__device__ __forceinline__ void avg4(const half_t* x, long row, int ldx, float (&t)[8]) {   // 2x2 average at (y, x): rows y, y + 1
  const half8 a = *(const half8*)x, b = *(const half8*)(x + ldx), c = *(const half8*)(x + row), d = *(const half8*)(x + row + ldx);
#pragma unroll
  for (int j = 0; j < 8; ++j) t[j] = ((((float)a[j] + (float)b[j]) + (float)c[j]) + (float)d[j]) * 0.25f;
}

// x (B,H,W,>=2c) -> p1 (B,H-1,W-1,c) = 2x2/s1 average of channels [0,c);  p2 (B,Ho,Wo,c) = 3x3/s2/p1 max of the averages of
// channels [c,2c), Ho = (H-2)/2+1;  arg (B,Ho,Wo,c) uint8: window position 3*ky+kx of the maximum.  Work items: the p1 elements
// first, then the p2 elements.
__global__ __launch_bounds__(256) void adown_fwd_kernel(const half_t* x, long x_bs, int ldx, half_t* p1, long p1_bs, int ld1, half_t* p2,
                                                        long p2_bs, int ld2, unsigned char* arg, int H, int W, int Ho, int Wo, int c,
                                                        long n1, long total) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const int cg = c / 8;
  const long row = (long)W * ldx;
  if (e < n1) {
    const int c8 = (int)(e % cg) * 8;
    long r = e / cg;
    const int xx = (int)(r % (W - 1));
    r /= (W - 1);
    const int yy = (int)(r % (H - 1));
    const long b = r / (H - 1);
    float t[8];
    avg4(x + b * x_bs + ((long)yy * W + xx) * ldx + c8, row, ldx, t);
    half8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = m355_to_half(t[j]);
    *(half8*)(p1 + b * p1_bs + ((long)yy * (W - 1) + xx) * ld1 + c8) = o;
    return;
  }
  const long e2 = e - n1;
  const int c8 = (int)(e2 % cg) * 8;
  long r = e2 / cg;
  const int xo = (int)(r % Wo);
  r /= Wo;
  const int yo = (int)(r % Ho);
  const long b = r / Ho;
  float best[8];
  int bi[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { best[j] = -INFINITY; bi[j] = 0; }
  bool first = true;
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const int yy = 2 * yo - 1 + ky;
    if (yy < 0 || yy >= H - 1) continue;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int xx = 2 * xo - 1 + kx;
      if (xx < 0 || xx >= W - 1) continue;
      float t[8];
      avg4(x + b * x_bs + ((long)yy * W + xx) * ldx + c + c8, row, ldx, t);
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (first || t[j] > best[j]) { best[j] = t[j]; bi[j] = 3 * ky + kx; }
      first = false;
    }
  }
  half8 o;
  unsigned long long packed = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    o[j] = m355_to_half(best[j]);
    packed |= (unsigned long long)bi[j] << (8 * j);
  }
  const long po = ((long)yo * Wo + xo);
  *(half8*)(p2 + b * p2_bs + po * ld2 + c8) = o;
  *(unsigned long long*)(arg + ((b * Ho * Wo + po) * c + c8)) = packed;
}

// gx (B,H,W,>=2c): channels [0,c): 0.25 x the sum of the <= 4 gradients g1 of the averages that contain the pixel; channels [c,2c):
// the same over ga, where ga(y', x') = sum of g2 over the <= 4 pooling windows whose stored argmax is (y', x') -- gathered window by
// window in ascending (yo, xo) order.  accumulate: gx += fp16(result) (fp32 add, one rounding: the arithmetic of Tensor.add_).
__global__ __launch_bounds__(256) void adown_bwd_kernel(const half_t* g1, long g1_bs, int ld1, const half_t* g2, long g2_bs, int ld2,
                                                        const unsigned char* arg, half_t* gx, long gx_bs, int ldg, int H, int W, int Ho, int Wo,
                                                        int c, int accumulate, long total) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const int cg2 = 2 * c / 8;
  const int c8 = (int)(e % cg2) * 8;
  long r = e / cg2;
  const int xx = (int)(r % W);
  r /= W;
  const int yy = (int)(r % H);
  const long b = r / H;
  float sum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int dy = 1; dy >= 0; --dy) {           // averages (y', x') = (yy - dy, xx - dx) in ascending order
    const int ya = yy - dy;
    if (ya < 0 || ya >= H - 1) continue;
#pragma unroll
    for (int dx = 1; dx >= 0; --dx) {
      const int xa = xx - dx;
      if (xa < 0 || xa >= W - 1) continue;
      if (c8 < c) {
        const half8 g = *(const half8*)(g1 + b * g1_bs + ((long)ya * (W - 1) + xa) * ld1 + c8);
#pragma unroll
        for (int j = 0; j < 8; ++j) sum[j] += (float)g[j];
      } else {
        // pooling windows that contain (ya, xa): yo with 2 yo - 1 <= ya <= 2 yo + 1
        const int yo0 = ya >> 1, yo1 = (ya + 1) >> 1, xo0 = xa >> 1, xo1 = (xa + 1) >> 1;
        for (int yo = yo0; yo <= yo1; ++yo) {
          if (yo >= Ho) continue;
          for (int xo = xo0; xo <= xo1; ++xo) {
            if (xo >= Wo) continue;
            const int pos = 3 * (ya - (2 * yo - 1)) + (xa - (2 * xo - 1));
            const long po = (long)yo * Wo + xo;
            const unsigned long long id = *(const unsigned long long*)(arg + ((b * Ho * Wo + po) * c + (c8 - c)));
            const half8 g = *(const half8*)(g2 + b * g2_bs + po * ld2 + (c8 - c));
#pragma unroll
            for (int j = 0; j < 8; ++j)
              if ((int)((id >> (8 * j)) & 0xff) == pos) sum[j] += (float)g[j];
          }
        }
      }
    }
  }
  half_t* const gp = gx + b * gx_bs + ((long)yy * W + xx) * ldg + c8;
  half8 o;
  if (accumulate) {
    const half8 old = *(const half8*)gp;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = m355_to_half((float)old[j] + (float)m355_to_half(sum[j] * 0.25f));
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = m355_to_half(sum[j] * 0.25f);
  }
  *(half8*)gp = o;
}

}  // namespace

int launch_addsilu_fwd(const half_t* a, const half_t* b, half_t* v, half_t* y, long npix, int ldy, int C, hipStream_t s) {
  if (!a || !b || !v || !y || npix < 1 || C < 8 || C % 8 || ldy % 8 || ldy < C) return -1;
  const long total = npix * (C / 8);
  hipLaunchKernelGGL(addsilu_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a, b, v, y, npix, ldy, C, total);
  return (int)hipGetLastError();
}

int launch_addsilu_bwd(const half_t* v, const half_t* dy, int lddy, half_t* g, long npix, int C, hipStream_t s) {
  if (!v || !dy || !g || npix < 1 || C < 8 || C % 8 || lddy % 8 || lddy < C) return -1;
  const long total = npix * (C / 8);
  hipLaunchKernelGGL(addsilu_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, v, dy, lddy, g, C, total);
  return (int)hipGetLastError();
}

int launch_adown_fwd(const half_t* x, long x_bs, int ldx, half_t* p1, long p1_bs, int ld1, half_t* p2, long p2_bs, int ld2, unsigned char* arg,
                     int B, int H, int W, int c, hipStream_t s) {
  if (!x || !p1 || !p2 || !arg || B < 1 || H < 2 || W < 2 || c < 8 || c % 8 || ldx % 8 || ld1 % 8 || ld2 % 8 || ldx < 2 * c) return -1;
  const int Ho = (H - 2) / 2 + 1, Wo = (W - 2) / 2 + 1;
  const long n1 = (long)B * (H - 1) * (W - 1) * (c / 8), total = n1 + (long)B * Ho * Wo * (c / 8);
  hipLaunchKernelGGL(adown_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, x_bs, ldx, p1, p1_bs, ld1, p2, p2_bs, ld2, arg,
                     H, W, Ho, Wo, c, n1, total);
  return (int)hipGetLastError();
}

int launch_adown_bwd(const half_t* g1, long g1_bs, int ld1, const half_t* g2, long g2_bs, int ld2, const unsigned char* arg, half_t* gx,
                     long gx_bs, int ldg, int B, int H, int W, int c, int accumulate, hipStream_t s) {
  if (!g1 || !g2 || !arg || !gx || B < 1 || H < 2 || W < 2 || c < 8 || c % 8 || ld1 % 8 || ld2 % 8 || ldg % 8 || ldg < 2 * c) return -1;
  const int Ho = (H - 2) / 2 + 1, Wo = (W - 2) / 2 + 1;
  const long total = (long)B * H * W * (2 * c / 8);
  hipLaunchKernelGGL(adown_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, g1, g1_bs, ld1, g2, g2_bs, ld2, arg, gx, gx_bs, ldg,
                     H, W, Ho, Wo, c, accumulate, total);
  return (int)hipGetLastError();
}

long colsum_workspace_floats(long nb, int cols) { return WS_HEAD + (nb > CS_BLOCKS ? nb : (long)CS_BLOCKS) * cols; }

// out[c] = sum over (b, r) of src[b * bstride + r * ld + c], c < cols, in a fixed order (partial rows + bn_finalize_kernel).
int launch_colsum(const void* src, int src_f16, long nb, long bstride, long rows, int ld, int cols, float* ws, float* out,
                  hipStream_t s) {
  if (!src || !ws || !out || nb < 1 || rows < 1 || cols < 1 || ld < cols) return -1;
  if (src_f16 ? (cols % 8 || ld % 8 || cols > 2048) : cols > 256) return -1;
  const int rl = src_f16 ? 256 / (cols / 8) : 256 / cols;
  long bpb = CS_BLOCKS / nb;                                 // blocks per batch entry: <= CS_BLOCKS rows in all (or nb of them)
  const long most = (rows + 8L * rl - 1) / (8L * rl);        // at least eight rows per lane
  if (bpb > most) bpb = most;
  if (bpb < 1) bpb = 1;
  const long chunk = (rows + bpb - 1) / bpb;
  bpb = (rows + chunk - 1) / chunk;
  if (src_f16)
    hipLaunchKernelGGL(colsum_f16_kernel, dim3((unsigned)bpb, (unsigned)nb), dim3(256), 0, s, (const half_t*)src, bstride, rows, ld, cols,
                       chunk, ws);
  else
    hipLaunchKernelGGL(colsum_f32_kernel, dim3((unsigned)bpb, (unsigned)nb), dim3(256), 0, s, (const float*)src, bstride, rows, ld, cols,
                       chunk, ws);
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((cols + 63) / 64), dim3(FIN_WAVES * 64), 0, s, ws, cols, (int)(bpb * nb), out);
  return (int)hipGetLastError();
}

int launch_upsample2x_bwd(const half_t* g, long g_bs, int ldg, half_t* d, long d_bs, int ldd, int B, int H, int W, int C, int accumulate,
                          hipStream_t s) {
  if (!g || !d || C % 8 || ldg % 8 || ldd % 8 || B < 1 || H < 1 || W < 1) return -1;
  const long total = (long)B * H * W * (C / 8);
  hipLaunchKernelGGL(upsample2x_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, g, g_bs, ldg, d, d_bs, ldd, H, W, C,
                     accumulate, total);
  return (int)hipGetLastError();
}

int launch_u8_to_f16x8(const unsigned char* src, half_t* dst, long npx, hipStream_t s) {
  if (!src || !dst || npx < 1) return -1;
  hipLaunchKernelGGL(u8_to_f16x8_kernel, dim3((unsigned)((npx + 1023) / 1024)), dim3(256), 0, s, src, dst, npx,
                     (int)(((unsigned long)src & 3) == 0));
  return (int)hipGetLastError();
}

}  // namespace m355
