// Does fp32 MFMA (32x32x2) overlap with VALU / LDS issue from the same or another wave on a SIMD?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NV, int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, float s) {
  f32x16 acc[NACC];
  for (int a = 0; a < NACC; ++a) for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
  float x = threadIdx.x * 1e-3f, y = s;
  float v[8]; for (int i = 0; i < 8; ++i) v[i] = x + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int a = 0; a < NACC; ++a) {
      acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[a], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < NV; ++j) v[j & 7] = __builtin_fmaf(v[j & 7], s, 1.0f);
    }
  }
  float r = 0; for (int a = 0; a < NACC; ++a) r += acc[a][0]; for (int i = 0; i < 8; ++i) r += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int NV, int NACC> float run(int blocks, int iters) {
  float* d; hipMalloc(&d, blocks * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<NV, NACC><<<blocks, 256>>>(d, iters, 1.0001f);
  hipEventRecord(e0); k<NV, NACC><<<blocks, 256>>>(d, iters, 1.0001f); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); hipFree(d); return ms;
}
int main() {
  const int iters = 20000;
  for (int occ = 1; occ <= 2; ++occ) {
    int blocks = 256 * occ;  // one or two waves per SIMD
    float base = run<0, 1>(blocks, iters);
    double tf = 2.0 * 32 * 32 * 2 * (double)iters * blocks * 4 / (base * 1e-3) / 1e12;
    printf("waves/SIMD %d: MFMA only (1 acc) %.3f ms = %.1f TF/s\n", occ, base, tf);
    printf("   +2 VALU/MFMA %.3f ms   +4 %.3f   +8 %.3f   +12 %.3f   +16 %.3f   +24 %.3f\n", run<2, 1>(blocks, iters),
           run<4, 1>(blocks, iters), run<8, 1>(blocks, iters), run<12, 1>(blocks, iters), run<16, 1>(blocks, iters), run<24, 1>(blocks, iters));
    printf("   2 acc: MFMA only %.3f   +8 VALU/MFMA %.3f  +16 %.3f\n", run<0, 2>(blocks, iters) / 2, run<8, 2>(blocks, iters) / 2, run<16, 2>(blocks, iters) / 2);
  }
  return 0;
}
