// Probe: semantics of ds_read_b64_tr_b16 (gfx950).  LDS holds s[row][col] = row*100 + col (fp16-exact small ints).
// Lane 4q+p of each 16-lane group passes the address of (row q, cols 4p..4p+3); prints what every lane receives.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short short4v __attribute__((ext_vector_type(4)));
__global__ void k(float* out) {
  __shared__ __attribute__((aligned(16))) _Float16 s[16 * 64];
  for (int i = threadIdx.x; i < 16 * 64; i += 64) s[i] = (_Float16)(float)((i / 64) * 100 + (i % 64));
  __syncthreads();
  const int lane = threadIdx.x, grp = lane >> 4, li = lane & 15, q = li >> 2, p = li & 3;
  auto* ptr = (__attribute__((address_space(3))) short4v*)(s + (grp * 4 + q) * 64 + p * 4);
  short4v v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(ptr);
  for (int j = 0; j < 4; ++j) out[lane * 4 + j] = (float)((_Float16*)&v)[j];
}
int main() {
  float* d; hipMalloc(&d, 256 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  float h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) printf("lane %2d: %6.0f %6.0f %6.0f %6.0f\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
  return 0;
}
