// Sustained fp32-MFMA rate and shader clock of an MI355X under (a) constant and (b) random operands.
// The 157.3 TFLOP/s fp32 matrix peak assumes 2.4 GHz; this measures what the part sustains for ~100 ms.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("hip error %d line %d\n", (int)e_, __LINE__); exit(1); } } while (0)

template <int NACC>
__global__ __launch_bounds__(256) void k(const float* in, float* out, long long* clk, int iters) {
  f32x16 acc[NACC];
  for (int a = 0; a < NACC; ++a) for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
  float x[8], y[8];
  for (int i = 0; i < 8; ++i) { x[i] = in[(threadIdx.x * 16 + i) & 4095]; y[i] = in[(threadIdx.x * 16 + 8 + i + blockIdx.x) & 4095]; }
  const long long c0 = __builtin_readcyclecounter();          // s_memtime
  const long long r0 = __builtin_amdgcn_s_memrealtime();       // constant 100 MHz
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(x[i], y[(i + a) & 7], acc[a], 0, 0, 0);
  }
  const long long c1 = __builtin_readcyclecounter();
  const long long r1 = __builtin_amdgcn_s_memrealtime();
  float r = 0; for (int a = 0; a < NACC; ++a) for (int e = 0; e < 16; ++e) r += acc[a][e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = r1 - r0; }
}

template <int NACC> void run(const char* name, const float* din, int blocks, int iters) {
  float* d; long long* c; CK(hipMalloc(&d, blocks * 256 * 4)); CK(hipMalloc(&c, 16));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  k<NACC><<<blocks, 256>>>(din, d, c, 1000);
  CK(hipEventRecord(e0)); k<NACC><<<blocks, 256>>>(din, d, c, iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  long long h[2]; CK(hipMemcpy(h, c, 16, hipMemcpyDeviceToHost));
  const double tf = 2.0 * 32 * 32 * 2 * 8.0 * NACC * (double)iters * blocks * 4 / (ms * 1e-3) / 1e12;
  printf("%-28s blocks %4d acc %d: %8.2f ms  %6.1f TFLOP/s   s_memtime/s_memrealtime = %.3f (x100 MHz)\n", name, blocks, NACC, ms, tf,
         (double)h[0] / (double)h[1]);
  CK(hipFree(d)); CK(hipFree(c));
}

int main() {
  std::vector<float> h(4096);
  float *dc, *dr, *dz;
  CK(hipMalloc(&dc, 4096 * 4)); CK(hipMalloc(&dr, 4096 * 4)); CK(hipMalloc(&dz, 4096 * 4));
  for (auto& v : h) v = 1.0f;
  CK(hipMemcpy(dc, h.data(), 4096 * 4, hipMemcpyHostToDevice));
  srand(1);
  for (auto& v : h) v = (rand() / (float)RAND_MAX - 0.5f) * 2.f;
  CK(hipMemcpy(dr, h.data(), 4096 * 4, hipMemcpyHostToDevice));
  for (auto& v : h) v = (rand() & 1) ? 0.f : (rand() / (float)RAND_MAX);   // post-ReLU like: half zeros
  CK(hipMemcpy(dz, h.data(), 4096 * 4, hipMemcpyHostToDevice));
  for (int rep = 0; rep < 2; ++rep) {
    run<1>("constant operands", dc, 512, 100000);
    run<1>("random operands", dr, 512, 100000);
    run<1>("half-zero (ReLU) operands", dz, 512, 100000);
    run<4>("random operands", dr, 512, 25000);
    run<4>("random operands, 1 wave/SIMD", dr, 256, 50000);
  }
  return 0;
}
