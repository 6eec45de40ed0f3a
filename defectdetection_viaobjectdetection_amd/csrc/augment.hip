// Training-time augmentation on the GPU (SURVEY.md A14 defaults / next row N2): 4-image mosaic, random affine
// (scale + translate), HSV gains and left-right flip fused into ONE gather kernel over the letterboxed uint8 image
// cache that lives in HBM.  Replaces upstream's CPU Mosaic -> RandomPerspective (cv2.warpAffine, bilinear, border
// 114) -> RandomHSV -> RandomFlip chain that runs in dataloader workers (reached from model.train(...),
// /root/reference/BscanBased/yolo_seg_train.py:12).  HBM-bound: per output pixel 4 neighbouring source texels of
// 3 bytes in, 3 bytes out.
//
// Geometry: a 2H x 2W canvas filled with 114 holds four H x W sources around the mosaic centre (xc, yc):
//   image 0 occupies [xc-W, xc) x [yc-H, yc), 1: [xc, xc+W) x [yc-H, yc), 2: [xc-W, xc) x [yc, yc+H), 3: [xc, xc+W) x [yc, yc+H)
// (each clipped by the canvas).  Output pixel (x, y) -- after the optional flip -- samples the canvas at
// minv * (x, y, 1) bilinearly (the inverse of the label transform the host applies to the polygons).
// With mosaic off, image 0 is placed at the canvas origin and minv maps into it directly.
#include "common.h"

namespace m355 {
namespace {

struct AugParams {      // one per output image (device array)
  int src[4];           // indices into the image cache
  float xc, yc;         // mosaic centre in canvas pixels
  float minv[6];        // canvas = minv * (x, y, 1): [a b c; d e f]
  float hgain, sgain, vgain;   // multiplicative gains on H (fraction of the hue circle), S, V
  int flip;             // 1: output column x reads column W-1-x
  int mosaic;           // 0: single image (src[0]) at the canvas origin
};

__device__ __forceinline__ void fetch(const uint8_t* cache, const AugParams& p, int H, int W, int cx, int cy, float* rgb) {
  // canvas texel (cx, cy) -> source image / texel, or the 114 border
  int img = -1, sx = 0, sy = 0;
  if (!p.mosaic) {
    if ((unsigned)cx < (unsigned)W && (unsigned)cy < (unsigned)H) { img = p.src[0]; sx = cx; sy = cy; }
  } else if ((unsigned)cx < (unsigned)(2 * W) && (unsigned)cy < (unsigned)(2 * H)) {
    const int xc = (int)p.xc, yc = (int)p.yc;
    const int right = cx >= xc, down = cy >= yc;
    sx = right ? cx - xc : cx - (xc - W);
    sy = down ? cy - yc : cy - (yc - H);
    if ((unsigned)sx < (unsigned)W && (unsigned)sy < (unsigned)H) img = p.src[down * 2 + right];
  }
  if (img < 0) {
    rgb[0] = rgb[1] = rgb[2] = 114.f;
    return;
  }
  const uint8_t* t = cache + (((long)img * H + sy) * W + sx) * 3;
  rgb[0] = (float)t[0]; rgb[1] = (float)t[1]; rgb[2] = (float)t[2];
}

__global__ __launch_bounds__(256) void augment_kernel(const uint8_t* cache, const AugParams* params, uint8_t* out, int B,
                                                      int H, int W) {
  const long total = (long)B * H * W;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int x = (int)(i % W);
    const long r = i / W;
    const int y = (int)(r % H);
    const int b = (int)(r / H);
    const AugParams p = params[b];
    const float xs = (float)(p.flip ? W - 1 - x : x), ys = (float)y;
    const float u = p.minv[0] * xs + p.minv[1] * ys + p.minv[2];
    const float v = p.minv[3] * xs + p.minv[4] * ys + p.minv[5];
    const float fu = floorf(u), fv = floorf(v);
    const int x0 = (int)fu, y0 = (int)fv;
    const float ax = u - fu, ay = v - fv;
    float c00[3], c01[3], c10[3], c11[3], rgb[3];
    fetch(cache, p, H, W, x0, y0, c00);
    fetch(cache, p, H, W, x0 + 1, y0, c01);
    fetch(cache, p, H, W, x0, y0 + 1, c10);
    fetch(cache, p, H, W, x0 + 1, y0 + 1, c11);
#pragma unroll
    for (int c = 0; c < 3; ++c)
      rgb[c] = (c00[c] * (1.f - ax) + c01[c] * ax) * (1.f - ay) + (c10[c] * (1.f - ax) + c11[c] * ax) * ay;
    if (p.hgain != 1.f || p.sgain != 1.f || p.vgain != 1.f) {   // RGB -> HSV, gains, -> RGB (V in 0..255)
      const float mx = fmaxf(rgb[0], fmaxf(rgb[1], rgb[2])), mn = fminf(rgb[0], fminf(rgb[1], rgb[2]));
      const float d = mx - mn;
      float h = 0.f;
      if (d > 0.f) {
        if (mx == rgb[0]) h = (rgb[1] - rgb[2]) / d;
        else if (mx == rgb[1]) h = 2.f + (rgb[2] - rgb[0]) / d;
        else h = 4.f + (rgb[0] - rgb[1]) / d;
        h *= (1.f / 6.f);
        if (h < 0.f) h += 1.f;
      }
      float s = mx > 0.f ? d / mx : 0.f;
      h = h * p.hgain;
      h -= floorf(h);
      s = fminf(s * p.sgain, 1.f);
      const float val = fminf(mx * p.vgain, 255.f);
      const float hh = h * 6.f;
      const int sector = (int)hh;
      const float f = hh - (float)sector;
      const float pq = val * (1.f - s), q = val * (1.f - s * f), t = val * (1.f - s * (1.f - f));
      switch (sector % 6) {
        case 0: rgb[0] = val; rgb[1] = t; rgb[2] = pq; break;
        case 1: rgb[0] = q; rgb[1] = val; rgb[2] = pq; break;
        case 2: rgb[0] = pq; rgb[1] = val; rgb[2] = t; break;
        case 3: rgb[0] = pq; rgb[1] = q; rgb[2] = val; break;
        case 4: rgb[0] = t; rgb[1] = pq; rgb[2] = val; break;
        default: rgb[0] = val; rgb[1] = pq; rgb[2] = q; break;
      }
    }
    uint8_t* o = out + i * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) o[c] = (uint8_t)fminf(fmaxf(floorf(rgb[c] + 0.5f), 0.f), 255.f);
  }
}

}  // namespace

int launch_augment(const uint8_t* cache, const void* params, uint8_t* out, int B, int H, int W, hipStream_t s) {
  if (B < 1 || H < 1 || W < 1) return -1;
  long blocks = ((long)B * H * W + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(augment_kernel, dim3((unsigned)blocks), dim3(256), 0, s, cache, (const AugParams*)params, out, B, H, W);
  return (int)hipGetLastError();
}

}  // namespace m355
