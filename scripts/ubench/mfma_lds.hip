// Inner loop of the implicit-GEMM <1,1> tile without global loads: how much of the LDS-fragment-read /
// barrier / ds_write time hides behind other waves' MFMAs on the same SIMD?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("hip error %d line %d\n", (int)e_, __LINE__); exit(1); } } while (0)
#define LDT 36

// MODE bits: 1 = LDS fragment reads, 2 = barriers, 4 = ds_write of the next tile, 8 = two accumulators per wave (<1,2>)
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int ksteps) {
  __shared__ __attribute__((aligned(16))) float lds[(128 + 64) * LDT];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lrow = lane & 31, lhalf = lane >> 5;
  constexpr int NT = (MODE & 8) ? 2 : 1;
  for (int i = tid; i < (128 + 64) * LDT; i += 256) lds[i] = (float)(i & 15) * 0.01f;
  __syncthreads();
  f32x16 acc[NT];
  for (int j = 0; j < NT; ++j) for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
  const float* As = lds; const float* Bs = lds + 128 * LDT;
  float4 ra[4], rb[NT];
  for (int i = 0; i < 4; ++i) ra[i] = make_float4(tid, i, 1.f, 2.f);
  for (int i = 0; i < NT; ++i) rb[i] = make_float4(tid, i, 3.f, 4.f);
  const int c4 = tid & 7, r0 = tid >> 3;
  float4 fa[2], fb[2][NT];
  fa[0] = fa[1] = make_float4(1.f, 2.f, 3.f, 4.f);
  for (int j = 0; j < NT; ++j) fb[0][j] = fb[1][j] = make_float4(1.f, 2.f, 3.f, 4.f);
  for (int ks = 0; ks < ksteps; ++ks) {
    auto frag = [&](int buf, int kg) {
      fa[buf] = *reinterpret_cast<const float4*>(&As[(wave * 32 + lrow) * LDT + kg * 8 + lhalf * 4]);
      for (int j = 0; j < NT; ++j) fb[buf][j] = *reinterpret_cast<const float4*>(&Bs[(j * 32 + lrow) * LDT + kg * 8 + lhalf * 4]);
    };
    if (MODE & 1) frag(0, 0);
#pragma unroll
    for (int kg = 0; kg < 4; ++kg) {
      if ((MODE & 1) && kg < 3) frag((kg + 1) & 1, kg + 1);
      const float4 a = fa[kg & 1];
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const float4 b = fb[kg & 1][j];
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc[j], 0, 0, 0);
      }
    }
    if (MODE & 2) __syncthreads();
    if (MODE & 4) {
#pragma unroll
      for (int i = 0; i < 4; ++i) *reinterpret_cast<float4*>(&lds[(r0 + 32 * i) * LDT + c4 * 4]) = ra[i];
#pragma unroll
      for (int i = 0; i < NT; ++i) *reinterpret_cast<float4*>(&lds[(128 + r0 + 32 * i) * LDT + c4 * 4]) = rb[i];
    }
    if (MODE & 2) __syncthreads();
  }
  float r = 0; for (int j = 0; j < NT; ++j) for (int e = 0; e < 16; ++e) r += acc[j][e];
  out[blockIdx.x * 256 + tid] = r;
}

template <int MODE> void run(const char* name, int occ, int ksteps) {
  const int blocks = 256 * occ;
  float* d; CK(hipMalloc(&d, (size_t)blocks * 256 * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  k<MODE><<<blocks, 256>>>(d, 100);
  CK(hipEventRecord(e0)); k<MODE><<<blocks, 256>>>(d, ksteps); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const int nt = (MODE & 8) ? 2 : 1;
  const double tf = 2.0 * 32 * 32 * 2 * 16.0 * nt * (double)ksteps * blocks * 4 / (ms * 1e-3) / 1e12;
  printf("  %-44s %d blocks/CU: %7.2f ms  %6.1f TFLOP/s\n", name, occ, ms, tf);
  CK(hipFree(d));
}

int main() {
  for (int occ = 1; occ <= 4; ++occ) {
    const int ks = 40000 / occ;
    run<0>("MFMA only", occ, ks);
    run<1>("+ LDS fragment reads", occ, ks);
    run<3>("+ reads + 2 barriers", occ, ks);
    run<7>("+ reads + barriers + ds_write (full loop)", occ, ks);
    run<15>("<1,2>: full loop", occ, ks);
  }
  return 0;
}
