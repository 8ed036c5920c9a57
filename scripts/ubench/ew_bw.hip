// Achievable read+write bandwidth of a streaming elementwise kernel (y -> z, 16 B per lane) as a function of the
// working-set size (L2 / Infinity Cache / HBM) and of the loads in flight per thread.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("hip error %d line %d\n", (int)e_, __LINE__); exit(1); } } while (0)

template <int U>
__global__ __launch_bounds__(256) void ew(const float4* __restrict__ in, float4* __restrict__ out, size_t n4, float s) {
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + (U - 1) * stride < n4; i += U * stride) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = in[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      v[u].x = fmaxf(v[u].x * s + 1.f, 0.f); v[u].y = fmaxf(v[u].y * s + 1.f, 0.f);
      v[u].z = fmaxf(v[u].z * s + 1.f, 0.f); v[u].w = fmaxf(v[u].w * s + 1.f, 0.f);
      out[i + u * stride] = v[u];
    }
  }
  for (; i < n4; i += stride) { float4 v = in[i]; v.x = fmaxf(v.x * s + 1.f, 0.f); out[i] = v; }
}

template <int U> void run(size_t mb, int blocks) {
  const size_t n4 = mb * 1024 * 1024 / 16;
  float4 *a, *b; CK(hipMalloc(&a, n4 * 16)); CK(hipMalloc(&b, n4 * 16));
  CK(hipMemset(a, 0, n4 * 16));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) ew<U><<<blocks, 256>>>(a, b, n4, 0.5f);
  CK(hipEventRecord(e0));
  const int reps = 20;
  for (int i = 0; i < reps; ++i) ew<U><<<blocks, 256>>>(a, b, n4, 0.5f);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("  %4zu MB in + %4zu MB out, unroll %d, %5d blocks: %7.1f us  %5.2f TB/s\n", mb, mb, U, blocks, 1e3 * ms / reps,
         2.0 * mb * 1.048576e6 / (ms / reps * 1e-3) / 1e12);
  CK(hipFree(a)); CK(hipFree(b));
}

int main() {
  for (size_t mb : {8, 29, 116, 1024}) {
    run<1>(mb, 4096); run<1>(mb, 16384); run<2>(mb, 4096); run<4>(mb, 2048); run<4>(mb, 4096); run<8>(mb, 2048);
  }
  return 0;
}
