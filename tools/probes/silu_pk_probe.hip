// Does the packed-fp32 form of the epilogue SiLU (v_pk_mul_f32 / v_pk_add_f32) give the same bits as the scalar form?
// hipcc -O3 --offload-arch=gfx950 silu_pk_probe.hip -o /tmp/silu_pk_probe && /tmp/silu_pk_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float silu1(float v) {
#pragma clang fp contract(off)
  const float e = __builtin_amdgcn_exp2f(v * -1.4426950408889634f);
  return v * __builtin_amdgcn_rcpf(1.0f + e);
}
__device__ __forceinline__ f2 silu2(f2 v) {
#pragma clang fp contract(off)
  f2 t = v * f2{-1.4426950408889634f, -1.4426950408889634f};
  t = f2{__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
  t = t + f2{1.0f, 1.0f};
  t = f2{__builtin_amdgcn_rcpf(t[0]), __builtin_amdgcn_rcpf(t[1])};
  return v * t;
}
__global__ void k(const float* x, float* a, float* b, float* c, float* d, int n, float s, float bias) {
  const int i = (blockIdx.x * 256 + threadIdx.x) * 2;
  if (i + 1 >= n) return;
  {
#pragma clang fp contract(off)
    a[i] = silu1(x[i]); a[i + 1] = silu1(x[i + 1]);
    const f2 r = silu2(f2{x[i], x[i + 1]});
    b[i] = r[0]; b[i + 1] = r[1];
    c[i] = x[i] * s + bias; c[i + 1] = x[i + 1] * s + bias;
    const f2 q = f2{x[i], x[i + 1]} * f2{s, s} + f2{bias, bias};
    d[i] = q[0]; d[i + 1] = q[1];
  }
}
int main() {
  const int n = 1 << 22;
  std::vector<float> h(n);
  srand(1);
  for (int i = 0; i < n; ++i) {
    const int m = i % 4;
    const float u = (float)rand() / RAND_MAX * 2.f - 1.f;
    h[i] = m == 0 ? u * 20.f : m == 1 ? u * 1e-3f : m == 2 ? u * 1e-30f : u * 100.f;
  }
  float *x, *a, *b, *c, *d;
  hipMalloc(&x, n * 4); hipMalloc(&a, n * 4); hipMalloc(&b, n * 4); hipMalloc(&c, n * 4); hipMalloc(&d, n * 4);
  hipMemcpy(x, h.data(), n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 512), dim3(256), 0, 0, x, a, b, c, d, n, 1.0f / 255.0f, 0.37f);
  std::vector<float> ha(n), hb(n), hc(n), hd(n);
  hipMemcpy(ha.data(), a, n * 4, hipMemcpyDeviceToHost); hipMemcpy(hb.data(), b, n * 4, hipMemcpyDeviceToHost);
  hipMemcpy(hc.data(), c, n * 4, hipMemcpyDeviceToHost); hipMemcpy(hd.data(), d, n * 4, hipMemcpyDeviceToHost);
  long ds = 0, dm = 0; int first = -1;
  for (int i = 0; i < n; ++i) {
    if (memcmp(&ha[i], &hb[i], 4)) { if (first < 0) first = i; ++ds; }
    if (memcmp(&hc[i], &hd[i], 4)) ++dm;
  }
  printf("silu: %ld of %d differ; mul+add: %ld differ\n", ds, n, dm);
  if (first >= 0) printf("first: x %.9g scalar %.9g packed %.9g\n", h[first], ha[first], hb[first]);
  return 0;
}
