// What does one VALU instruction of a given kind cost an fp32-MFMA loop on the same SIMD?  (The MFMA shares the VALU
// issue port; integer multiplies and 64-bit shifts are suspected to run at a quarter of the fp32 rate.)
// Each wave rotates 4 independent 32x32x2 accumulators and issues NV ops of one kind per MFMA; the added time per
// op is reported in SIMD cycles, taking one MFMA = 64 cycles from the MFMA-only run.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x16 __attribute__((ext_vector_type(16)));
enum { K_FMA, K_MULLO, K_MULHI, K_MAD64, K_MUL24, K_MAD24, K_CVT, K_SHR64, K_CNDMASK, K_ADD, K_NKIND };
static const char* kNames[] = {"v_fma_f32", "v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u64_u32", "v_mul_u32_u24",
                               "v_mad_u32_u24", "v_cvt_f32_u32", "v_lshrrev_b64", "v_cndmask_b32", "v_add_u32"};
template <int KIND>
__device__ __forceinline__ void op(uint32_t& a, uint32_t b, float& f, uint64_t& w) {
  // exactly ONE instruction per op (inline asm: nothing folds)
  if (KIND == K_FMA) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f) : "v"(b));
  if (KIND == K_MULLO) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a) : "v"(b));
  if (KIND == K_MULHI) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a) : "v"(b));
  if (KIND == K_MAD64) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w) : "v"(a), "v"(b) : "vcc");
  if (KIND == K_MUL24) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a) : "v"(b));
  if (KIND == K_MAD24) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(a) : "v"(b));
  if (KIND == K_CVT) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(a));
  if (KIND == K_SHR64) asm volatile("v_lshrrev_b64 %0, 1, %0" : "+v"(w));
  if (KIND == K_CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a) : "v"(b) : "vcc");
  if (KIND == K_ADD) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b));
}
template <int KIND, int NV>
__global__ __launch_bounds__(256) void k(float* out, int iters, uint32_t b) {
  f32x16 acc[4];
  for (int a = 0; a < 4; ++a)
    for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
  const float x = threadIdx.x * 1e-3f, y = 1.0001f;
  uint32_t v[4];
  float f[4];
  uint64_t w[4];
  for (int i = 0; i < 4; ++i) v[i] = threadIdx.x * 7 + i + 3, f[i] = x + i, w[i] = v[i] * 0x100000001ull;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[a], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < NV; ++j) op<KIND>(v[j & 3], b, f[j & 3], w[j & 3]);
    }
  }
  float r = 0;
  for (int a = 0; a < 4; ++a) r += acc[a][0];
  for (int i = 0; i < 4; ++i) r += f[i] + (float)v[i] + (float)w[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int KIND, int NV>
float run(int blocks, int iters) {
  float* d;
  hipMalloc(&d, blocks * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  k<KIND, NV><<<blocks, 256>>>(d, iters, 77u);
  hipEventRecord(e0);
  k<KIND, NV><<<blocks, 256>>>(d, iters, 77u);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  hipFree(d);
  return ms;
}
template <int KIND>
void kind(int blocks, int iters, float base) {
  const float t4 = run<KIND, 4>(blocks, iters), t8 = run<KIND, 8>(blocks, iters);
  // one MFMA = base / (4 * iters) per wave slot = 64 cycles (x waves per SIMD)
  const double cyc_per_ms = 64.0 * 4 * iters / base;
  printf("  %-28s +4/MFMA %.3f ms (%.1f cyc/op)   +8/MFMA %.3f ms (%.1f cyc/op)\n", kNames[KIND], t4,
         (t4 - base) * cyc_per_ms / (16.0 * iters), t8, (t8 - base) * cyc_per_ms / (32.0 * iters));
}
int main() {
  const int iters = 10000;
  for (int occ = 1; occ <= 2; ++occ) {
    const int blocks = 256 * occ;
    const float base = run<K_FMA, 0>(blocks, iters);
    printf("waves/SIMD %d: MFMA only %.3f ms (per-op cycles are per wave; x%d waves share the SIMD)\n", occ, base, occ);
    kind<K_FMA>(blocks, iters, base);
    kind<K_ADD>(blocks, iters, base);
    kind<K_CNDMASK>(blocks, iters, base);
    kind<K_MUL24>(blocks, iters, base);
    kind<K_MAD24>(blocks, iters, base);
    kind<K_CVT>(blocks, iters, base);
    kind<K_MULLO>(blocks, iters, base);
    kind<K_MULHI>(blocks, iters, base);
    kind<K_MAD64>(blocks, iters, base);
    kind<K_SHR64>(blocks, iters, base);
  }
  return 0;
}
