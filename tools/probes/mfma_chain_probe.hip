// Probe: cycles per v_mfma_f32_32x32x16_f16 for one wave per SIMD (256-thread block, one block per CU) in four streams:
//   A: 18-long dependent chains, accumulator after accumulator (pixel-block major)
//   B: the same MFMAs interleaved over 9 accumulators (dependent distance 9)
//   C: A + one ds_read_b128 per MFMA (FIFO of 8), D: B + the same reads.
// Build: hipcc --offload-arch=gfx950 -O3 mfma_chain_probe.hip -o mfma_chain_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(256, 1) void probe(unsigned long long* out, float* sink, int reps) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 8192; i += 256) ((float*)smem)[i] = (float)(i % 17) * 0.01f;
  __syncthreads();
  float16v acc[9];
  half8 a[18], bq[9];
  for (int k = 0; k < 9; ++k) for (int j = 0; j < 16; ++j) acc[k][j] = 0.f;
  for (int i = 0; i < 18; ++i) for (int j = 0; j < 8; ++j) a[i][j] = (_Float16)(0.001f * (i + j + lane));
  for (int i = 0; i < 9; ++i) bq[i] = *(const half8*)(smem + lane * 16 + i * 1024);
  const int row = lane & 31;
  const int base = row * 64 + (((lane >> 5) ^ ((row >> 2) & 3)) << 4);   // the conflict-free plane layout of conv3x3_planes.hip
  __builtin_amdgcn_s_barrier();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < reps; ++r) {
#pragma unroll
    for (int n = 0; n < 162; ++n) {
      const int k = (MODE & 1) ? n % 9 : n / 18;
      const int i = (MODE & 1) ? n / 9 : n % 18;
      acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], bq[n % 9], acc[k], 0, 0, 0);
      if (MODE & 2) bq[(n + 8) % 9] = *(const half8*)(smem + (base ^ ((n & 1) << 5)) + (n % 8) * 2048);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int k = 0; k < 9; ++k) for (int j = 0; j < 16; ++j) s += acc[k][j];
  sink[blockIdx.x * 256 + threadIdx.x] = s;
  if (lane == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int MODE>
void run(const char* name, unsigned long long* d_out, float* d_sink) {
  const int reps = 50;
  unsigned long long h[1024];
  for (int w = 0; w < 2; ++w) {
    hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(256), 65536, 0, d_out, d_sink, reps);
    hipDeviceSynchronize();
  }
  hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost);
  double sum = 0;
  for (int i = 0; i < 1024; ++i) sum += (double)h[i];
  printf("%s: %.2f cycles per MFMA (mean over 1024 waves)\n", name, sum / 1024 / (162.0 * reps));
}

int main() {
  unsigned long long* d_out; float* d_sink;
  hipMalloc(&d_out, 1024 * 8); hipMalloc(&d_sink, 256 * 256 * 4);
  hipFuncSetAttribute((const void*)probe<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  run<0>("A chain-of-18, no LDS      ", d_out, d_sink);
  run<1>("B interleaved x9, no LDS   ", d_out, d_sink);
  run<2>("C chain-of-18 + ds_read/MFMA", d_out, d_sink);
  run<3>("D interleaved + ds_read/MFMA", d_out, d_sink);
  return 0;
}
