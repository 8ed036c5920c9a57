// Does a SHORT fp32-MFMA kernel reach the sustained rate?  The same dependent-distance-4 MFMA loop as mfma_clock.hip
// (4 accumulators per wave) on several grids (blocks x threads: 1 .. 4 waves per SIMD, as one or several workgroups per
// CU), launched back to back in batches of 20 at several lengths: 15 us .. 13 ms per launch.  Prints TFLOP/s per launch
// length.  Measured (profiles/r03_ubench_mfma_short.txt): 85-92 % of the 2.4-GHz peak for 40-60 us launches, 91-96 % at
// 100-200 us, 98.3 % at 3-13 ms, for every balanced grid -- the clock ramp of mfma_ramp.hip, not occupancy.  (Built
// WITHOUT -amdgpu-mfma-vgpr-form and with __launch_bounds__(256) the kernel takes 68 VGPRs + 64 AGPRs = 3 waves per SIMD
// and 1024 x 256 threads need two rounds: 65 %.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("hip error %d line %d\n", (int)e_, __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(1024) void k(const float* in, float* out, int iters) {
  f32x16 acc[4];
  for (int a = 0; a < 4; ++a) for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
  float x[8], y[8];
  for (int i = 0; i < 8; ++i) { x[i] = in[(threadIdx.x * 16 + i) & 4095]; y[i] = in[(threadIdx.x * 16 + 8 + i + blockIdx.x) & 4095]; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int a = 0; a < 4; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(x[i], y[(i + a) & 7], acc[a], 0, 0, 0);
  }
  float r = 0; for (int a = 0; a < 4; ++a) for (int e = 0; e < 16; ++e) r += acc[a][e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

int main() {
  std::vector<float> h(4096);
  srand(1);
  for (auto& v : h) v = (rand() / (float)RAND_MAX - 0.5f) * 2.f;
  float *din, *d;
  CK(hipMalloc(&din, 4096 * 4)); CK(hipMalloc(&d, 4096 * 1024 * 4));
  CK(hipMemcpy(din, h.data(), 4096 * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 2; ++rep)
    for (int cfg : {768256, 1024256, 256256, 256512, 256768, 2561024, 512512, 512256, 384512}) {
      const int blocks = cfg / 1000, threads = cfg % 1000 == 24 ? 1024 : cfg % 1000;
      for (int iters : {15, 60, 480, 3840}) {
        for (int w = 0; w < 3; ++w) k<<<blocks, threads>>>(din, d, iters);
        CK(hipEventRecord(e0));
        for (int l = 0; l < 20; ++l) k<<<blocks, threads>>>(din, d, iters);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / 20;
        const double tf = 2.0 * 32 * 32 * 2 * 32.0 * (double)iters * blocks * (threads / 64) / (us * 1e-6) / 1e12;
        printf("blocks %4d x %4d threads  %5d x 32 MFMAs per wave: %9.1f us per launch  %6.1f TFLOP/s (%4.1f %%)\n", blocks, threads, iters, us, tf, tf / 1.573);
      }
    }
  return 0;
}
