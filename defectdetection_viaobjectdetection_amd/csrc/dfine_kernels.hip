// D-FINE decoder hot ops: the multi-scale deformable-attention core and the box decode that
// /root/reference/D-Fine/temporal_dfine.py:160-181 reaches through transformers' modeling_d_fine.py
// (multi_scale_deformable_attention_v2 :150-221, DFineIntegral :756-778, distance2bbox :1115-1137).
//
// msda: for every (batch, query, head) the weighted sum over P sampling points of a bilinear (or nearest, "discrete")
// sample of that head's 32-channel slice of the value map of the point's level.  A gather kernel: 4 corners x 128 B
// per point, 6 KB per (b, q, h); bound by L2 / Infinity-Cache gather bandwidth, no MFMA.  Layout as the reference
// passes it: value (B, S, heads, 32) fp32 -> a corner is one coalesced 128-byte read by 32 lanes.
#include "common.h"

namespace m355 {
namespace {

struct MsdaArgs {
  const float* value;   // (B, S, H, 32)
  const float* loc;     // (B, Q, H, P, 2), normalised [0, 1] ("default") or as given ("discrete")
  const float* attn;    // (B, Q, H, P)
  float* out;           // (B, Q, H * 32)
  int B, S, H, Q, P, L;
  int lh[8], lw[8], lstart[8], pend[8];   // level height / width / first pixel / one past its last point
  int discrete;
  // module mode (DFineMultiscaleDeformableAttention.forward with 4-d reference points, modeling_d_fine.py:268-296):
  // loc holds the RAW sampling offsets and attn the RAW attention logits of the two linear layers; the kernel forms
  //   location = ref.xy + offset * (1 / points of the level) * ref.wh * offset_scale,   weight = softmax over the P points
  const float* ref;     // (B, Q, 4) cx, cy, w, h; nullptr: loc / attn are final
  float offset_scale;
};

__global__ __launch_bounds__(256) void msda_kernel(const MsdaArgs a) {
  const int c = threadIdx.x & 31;
  const long triple = (long)blockIdx.x * 8 + (threadIdx.x >> 5);
  const long ntriples = (long)a.B * a.Q * a.H;
  if (triple >= ntriples) return;   // whole 32-lane groups leave together: the shuffles below stay inside a group
  const int h = (int)(triple % a.H);
  const long bq = triple / a.H;
  const int b = (int)(bq / a.Q);
  // lanes 0 .. P-1 of the group fetch the point data once, everybody reads it by shuffle
  float lx = 0.f, ly = 0.f, aw = 0.f;
  if (c < a.P) {
    const float* lp = a.loc + (triple * a.P + c) * 2;
    lx = lp[0];
    ly = lp[1];
    aw = a.attn[triple * a.P + c];
  }
  const float* vb = a.value + (long)b * a.S * a.H * 32 + h * 32 + c;
  const long pix_stride = (long)a.H * 32;
  float acc = 0.f;
  int p = 0;
  for (int l = 0; l < a.L; ++l) {
    const int W = a.lw[l], Hh = a.lh[l];
    const float* vl = vb + (long)a.lstart[l] * pix_stride;
    for (; p < a.pend[l]; ++p) {
      const float x = __shfl(lx, p, 32), y = __shfl(ly, p, 32), w = __shfl(aw, p, 32);
      float v;
      if (a.discrete) {
        // (loc * (W, H) + 0.5).to(int64) truncates toward zero, then clamps
        long xi = (long)(x * (float)W + 0.5f), yi = (long)(y * (float)Hh + 0.5f);
        xi = xi < 0 ? 0 : (xi > W - 1 ? W - 1 : xi);
        yi = yi < 0 ? 0 : (yi > Hh - 1 ? Hh - 1 : yi);
        v = vl[(yi * W + xi) * pix_stride];
      } else {
        // grid_sample(align_corners=False, padding_mode="zeros") on grid = 2 * loc - 1
        const float gx = 2.f * x - 1.f, gy = 2.f * y - 1.f;
        const float ix = ((gx + 1.f) * (float)W - 1.f) * 0.5f, iy = ((gy + 1.f) * (float)Hh - 1.f) * 0.5f;
        const float fx = floorf(ix), fy = floorf(iy);
        const float we = ix - fx, ws = iy - fy, ww = 1.f - we, wn = 1.f - ws;
        const int x0 = (int)fx, y0 = (int)fy;
        const bool xv0 = (unsigned)x0 < (unsigned)W, xv1 = (unsigned)(x0 + 1) < (unsigned)W;
        const bool yv0 = (unsigned)y0 < (unsigned)Hh, yv1 = (unsigned)(y0 + 1) < (unsigned)Hh;
        const float* r0 = vl + ((long)y0 * W + x0) * pix_stride;
        const float* r1 = r0 + (long)W * pix_stride;
        const float nw = (xv0 && yv0) ? r0[0] : 0.f, ne = (xv1 && yv0) ? r0[pix_stride] : 0.f;
        const float sw = (xv0 && yv1) ? r1[0] : 0.f, se = (xv1 && yv1) ? r1[pix_stride] : 0.f;
        v = nw * (ww * wn) + ne * (we * wn) + sw * (ww * ws) + se * (we * ws);
      }
      acc += v * w;
    }
  }
  a.out[triple * 32 + c] = acc;
}

// Wide form for P <= 16 (D-FINE: 12): one wave per (b, q, h).  Lane k < 4 P owns the (point k >> 2, corner k & 3)
// pair: it computes that corner's pixel offset and weight x attention once.  The wave then walks the pairs eight at a
// time: lane = (slot, c4) reads 16 bytes of corner `8 it + slot`, so ONE load instruction fetches eight 128-byte
// corners (the scalar form above needs 32 lanes x 4 B per corner and 48 dependent rounds per lane).  Partial sums are
// reduced across the eight slots with three xor-shuffles.
__global__ __launch_bounds__(256) void msda_wave_kernel(const MsdaArgs a) {
  const int lane = threadIdx.x & 63;
  const long triple = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (triple >= (long)a.B * a.Q * a.H) return;   // whole waves leave together
  const int h = (int)(triple % a.H);
  const int b = (int)(triple / a.H / a.Q);
  // module mode: softmax of the P raw logits.  Point p sits on lanes 4p .. 4p+3, so the xor-shuffles 4 .. 32 reduce over
  // the points inside each corner class; idle lanes carry -inf / 0.
  float soft = 0.f;
  if (a.ref) {
    const float z = lane < 4 * a.P ? a.attn[triple * a.P + (lane >> 2)] : -INFINITY;
    float m = z;
#pragma unroll
    for (int k = 4; k < 64; k <<= 1) m = fmaxf(m, __shfl_xor(m, k, 64));
    const float e = lane < 4 * a.P ? __expf(z - m) : 0.f;
    float sum = e;
#pragma unroll
    for (int k = 4; k < 64; k <<= 1) sum += __shfl_xor(sum, k, 64);
    soft = e / sum;
  }
  int my_off = 0;       // pixel index (level start included) of this lane's corner, 0 when the corner contributes nothing
  float my_wt = 0.f;
  if (lane < 4 * a.P) {
    const int p = lane >> 2, corner = lane & 3;
    int l = 0;
    while (p >= a.pend[l]) ++l;
    const int W = a.lw[l], Hh = a.lh[l];
    const float* lp = a.loc + (triple * a.P + p) * 2;
    float x = lp[0], y = lp[1], aw = a.ref ? soft : a.attn[triple * a.P + p];
    if (a.ref) {
      const float* rp = a.ref + (triple / a.H) * 4;
      const float nscale = 1.0f / (float)(a.pend[l] - (l ? a.pend[l - 1] : 0));
      x = rp[0] + x * nscale * rp[2] * a.offset_scale;   // reference op order: offsets * scale * wh * offset_scale
      y = rp[1] + y * nscale * rp[3] * a.offset_scale;
    }
    if (a.discrete) {
      long xi = (long)(x * (float)W + 0.5f), yi = (long)(y * (float)Hh + 0.5f);
      xi = xi < 0 ? 0 : (xi > W - 1 ? W - 1 : xi);
      yi = yi < 0 ? 0 : (yi > Hh - 1 ? Hh - 1 : yi);
      my_off = a.lstart[l] + (int)(yi * W + xi);
      my_wt = corner == 0 ? aw : 0.f;
    } else {
      const float gx = 2.f * x - 1.f, gy = 2.f * y - 1.f;
      const float ix = ((gx + 1.f) * (float)W - 1.f) * 0.5f, iy = ((gy + 1.f) * (float)Hh - 1.f) * 0.5f;
      const float fx = floorf(ix), fy = floorf(iy);
      const float we = ix - fx, ws = iy - fy;
      // floorf of a huge / non-finite coordinate: the int conversion saturates, the range test below rejects it
      const float cxf = fx + (float)(corner & 1), cyf = fy + (float)(corner >> 1);
      const bool ok = cxf >= 0.f && cxf <= (float)(W - 1) && cyf >= 0.f && cyf <= (float)(Hh - 1);
      const float wx = (corner & 1) ? we : 1.f - we, wy = (corner >> 1) ? ws : 1.f - ws;
      if (ok) {
        my_off = a.lstart[l] + (int)cyf * W + (int)cxf;
        my_wt = (wx * wy) * aw;
      }
    }
  }
  const int slot = lane >> 3, c4 = lane & 7;
  const float4* vb = (const float4*)(a.value + (long)b * a.S * a.H * 32 + h * 32) + c4;
  const long pix_stride4 = (long)a.H * 8;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  const int iters = (4 * a.P + 7) >> 3;
  for (int it = 0; it < iters; ++it) {
    const int k = it * 8 + slot;                       // k >= 4 P reads lanes that hold weight 0 / offset 0
    const int off = __shfl(my_off, k, 64);
    const float wt = __shfl(my_wt, k, 64);
    const float4 v = vb[(long)off * pix_stride4];
    acc.x += wt * v.x; acc.y += wt * v.y; acc.z += wt * v.z; acc.w += wt * v.w;
  }
#pragma unroll
  for (int m = 8; m < 64; m <<= 1) {
    acc.x += __shfl_xor(acc.x, m, 64); acc.y += __shfl_xor(acc.y, m, 64);
    acc.z += __shfl_xor(acc.z, m, 64); acc.w += __shfl_xor(acc.w, m, 64);
  }
  if (slot == 0) ((float4*)(a.out + triple * 32))[c4] = acc;
}

// One thread per box: softmax over the bins of each of the four sides, dot with the weighting function W(n),
// distance2bbox against the (cx, cy, w, h) reference point, corners -> centre format, optional clamp to [0, 1].
__global__ void dfine_decode_kernel(const float* dist, const float* project, const float* ref, float* boxes, long n,
                                    int nbins1, float reg_scale, int clamp01) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float d[4];
  for (int s = 0; s < 4; ++s) {
    const float* z = dist + (i * 4 + s) * nbins1;
    float m = z[0];
    for (int k = 1; k < nbins1; ++k) m = fmaxf(m, z[k]);
    float sum = 0.f, dot = 0.f;
    for (int k = 0; k < nbins1; ++k) {
      const float e = expf(z[k] - m);
      sum += e;
      dot += e * project[k];
    }
    d[s] = dot / sum;
  }
  const float rs = fabsf(reg_scale);
  const float cx = ref[i * 4], cy = ref[i * 4 + 1], w = ref[i * 4 + 2], hh = ref[i * 4 + 3];
  const float x0 = cx - (0.5f * rs + d[0]) * (w / rs), y0 = cy - (0.5f * rs + d[1]) * (hh / rs);
  const float x1 = cx + (0.5f * rs + d[2]) * (w / rs), y1 = cy + (0.5f * rs + d[3]) * (hh / rs);
  float o[4] = {(x0 + x1) / 2.f, (y0 + y1) / 2.f, x1 - x0, y1 - y0};
  for (int k = 0; k < 4; ++k) {
    float v = o[k];
    // torch.clamp keeps NaN (the reference feeds pre-sigmoid reference points, inf - inf happens); fminf / fmaxf would not
    if (clamp01 && v == v) v = v < 0.f ? 0.f : (v > 1.f ? 1.f : v);
    boxes[i * 4 + k] = v;
  }
}

}  // namespace

int launch_msda(const float* value, const float* loc, const float* attn, float* out, int B, int S, int H, int D, int Q,
                int P, int L, const int* shapes_hw, const int* points_per_level, int discrete, hipStream_t s, const float* ref,
                float offset_scale) {
  if (!value || !loc || !attn || !out || !shapes_hw || !points_per_level) return -1;
  if (ref && (P > 16 || discrete)) return -1;   // module mode lives in the wave kernel (bilinear, at most 16 points)
  if (D != 32 || L < 1 || L > 8 || P < 1 || P > 32 || B < 1 || Q < 1 || H < 1) return -1;
  MsdaArgs a{};
  a.value = value; a.loc = loc; a.attn = attn; a.out = out;
  a.B = B; a.S = S; a.H = H; a.Q = Q; a.P = P; a.L = L; a.discrete = discrete;
  a.ref = ref; a.offset_scale = offset_scale;
  int start = 0, pend = 0;
  for (int l = 0; l < L; ++l) {
    a.lh[l] = shapes_hw[2 * l]; a.lw[l] = shapes_hw[2 * l + 1];
    if (a.lh[l] < 1 || a.lw[l] < 1 || points_per_level[l] < 0) return -1;
    a.lstart[l] = start;
    start += a.lh[l] * a.lw[l];
    pend += points_per_level[l];
    a.pend[l] = pend;
  }
  if (start != S || pend != P) return -1;   // the level table must tile the value sequence and the point axis
  const long ntriples = (long)B * Q * H;
  if (P <= 16)
    hipLaunchKernelGGL(msda_wave_kernel, dim3((unsigned)((ntriples + 3) / 4)), dim3(256), 0, s, a);
  else
    hipLaunchKernelGGL(msda_kernel, dim3((unsigned)((ntriples + 7) / 8)), dim3(256), 0, s, a);
  return (int)hipGetLastError();
}

int launch_dfine_decode(const float* dist, const float* project, const float* ref, float* boxes, long n, int nbins1,
                        float reg_scale, int clamp01, hipStream_t s) {
  if (!dist || !project || !ref || !boxes || n < 0 || nbins1 < 2 || reg_scale == 0.f) return -1;
  if (n == 0) return 0;
  hipLaunchKernelGGL(dfine_decode_kernel, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, s, dist, project, ref, boxes, n,
                     nbins1, reg_scale, clamp01);
  return (int)hipGetLastError();
}

}  // namespace m355
