// Probe: what does `buffer_load_dwordx4 ... offen lds` do for lanes whose offset fails the buffer range check?
// (a) writes zeros to their LDS slot, or (b) leaves the slot untouched.  Build: hipcc --offload-arch=gfx950 -o probe probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(const float* src, int nbytes, float* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* f = (float*)smem;
  for (int i = threadIdx.x; i < 256; i += 64) f[i] = -7.0f;          // pattern
  __syncthreads();
  auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nbytes, 0x00020000);
  int voff = (threadIdx.x & 1) ? (int)0x80000000 : (int)(threadIdx.x * 16);  // odd lanes out of range
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)smem, 16, voff, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 256; i += 64) out[i] = f[i];
}
int main() {
  float *src, *out, h[256], hs[256];
  for (int i = 0; i < 256; ++i) hs[i] = 1.0f + i;
  hipMalloc(&src, 1024); hipMalloc(&out, 1024);
  hipMemcpy(src, hs, 1024, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 1024, 0, src, 1024, out);
  hipMemcpy(h, out, 1024, hipMemcpyDeviceToHost);
  printf("lane0 slot (in range): %g %g %g %g\n", h[0], h[1], h[2], h[3]);
  printf("lane1 slot (out of range): %g %g %g %g  -> %s\n", h[4], h[5], h[6], h[7],
         h[4] == 0.f ? "ZEROS WRITTEN" : (h[4] == -7.f ? "UNTOUCHED" : "OTHER"));
  printf("lane3 slot: %g, lane2 slot: %g\n", h[12], h[8]);
  return 0;
}
