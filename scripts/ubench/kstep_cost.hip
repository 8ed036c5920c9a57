// What in the K-step of the <1,1> conv tile keeps the loop at ~83 % of the MFMA peak in steady state?  (round 6)
// mfma_chain.hip showed: 16 MFMAs + the 8 fragment ds_read_b128 of a K-step run at 98 %.  This loop adds the rest of the K-step
// of conv_halo_kernel<1,1> one ingredient at a time, 4-wave workgroups, 1 ... 4 workgroups per CU:
//   W  the tile store of the next tap's weights (1 ds_write_b128 per lane into the other stage)
//   B  the workgroup barrier that publishes it (one s_barrier per K-step = per 16 MFMAs)
//   G  the global load that feeds the store (1 buffer_load_dwordx4 per lane per K-step from an L2-resident array, waited for
//      before the store)
//   B3 a barrier every THIRD K-step (= 48 MFMAs between barriers: three taps' weights per stage)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int W, int B, int G>
__global__ __launch_bounds__(256) void k(float* out, const float4* __restrict__ src, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[2 * (128 + 32) * 36];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 2 * (128 + 32) * 36; i += 256) lds[i] = 1e-3f * (float)(i & 31);
  __syncthreads();
  f32x16 acc;
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  const int lrow = lane & 31, lhalf = lane >> 5;
  const float* A0 = lds + (wave * 32 + lrow) * 36 + lhalf * 4;   // this wave's 32 A rows
  const float* B0 = lds + 128 * 36 + lrow * 36 + lhalf * 4;      // the shared 32 B rows
  float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
  const float4* sp = src + (blockIdx.x & 63) * 256 + tid;
  for (int it = 0; it < iters; ++it) {
    const int st = (it & 1) * (128 + 32) * 36;
    if (G) g = sp[(it & 7) * 64 * 256];
    float4 fa[2], fb[2];
    fa[0] = *reinterpret_cast<const float4*>(A0 + st);
    fb[0] = *reinterpret_cast<const float4*>(B0 + st);
    fa[1] = *reinterpret_cast<const float4*>(A0 + st + 8);
    fb[1] = *reinterpret_cast<const float4*>(B0 + st + 8);
#pragma unroll
    for (int gk = 0; gk < 4; ++gk) {
      const int buf = gk & 1;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[buf].x, fb[buf].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[buf].y, fb[buf].y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[buf].z, fb[buf].z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[buf].w, fb[buf].w, acc, 0, 0, 0);
      if (gk < 2) {
        fa[buf] = *reinterpret_cast<const float4*>(A0 + st + 8 * (gk + 2));
        fb[buf] = *reinterpret_cast<const float4*>(B0 + st + 8 * (gk + 2));
      }
    }
    if (G) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // the global load stays at the top of the K-step: a full step of cover
    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (W) {   // the next stage's B tile: 32 rows x 32 floats = 256 lanes x 16 B
      float4 v = g;
      v.x += (float)it;
      *reinterpret_cast<float4*>(lds + ((it + 1) & 1) * (128 + 32) * 36 + 128 * 36 + (tid >> 3) * 36 + (tid & 7) * 4) = v;
    }
    if (B == 1 || (B == 3 && it % 3 == 2)) __syncthreads();
  }
  float r = 0;
  for (int e = 0; e < 16; ++e) r += acc[e];
  out[blockIdx.x * 256 + tid] = r + g.x;
}
template <int W, int B, int G> double run(int blocks, int iters, const float4* src) {
  float* d;
  (void)hipMalloc(&d, (size_t)blocks * 256 * 4);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) k<W, B, G><<<blocks, 256>>>(d, src, iters);
  (void)hipEventRecord(e0);
  k<W, B, G><<<blocks, 256>>>(d, src, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipFree(d);
  return 2.0 * 32 * 32 * 2 * 16.0 * iters * blocks * 4 / (ms * 1e-3) / 1e12;
}
int main() {
  const int iters = 3000;
  float4* src;
  (void)hipMalloc(&src, (size_t)8 * 64 * 256 * sizeof(float4));     // 2 MB: L2 resident
  (void)hipMemset(src, 0, (size_t)8 * 64 * 256 * sizeof(float4));
  printf("TFLOP/s of a <1,1> K-step (16 MFMAs + 8 fragment reads per wave), peak 157.3\n");
  printf("WG/CU |  reads only | + W store | + W + barrier | + W + B + global load | + W + G, barrier every 3rd step\n");
  for (int occ = 1; occ <= 4; ++occ) {
    const int blocks = 256 * occ;
    printf("  %d   %9.1f %11.1f %13.1f %17.1f %21.1f\n", occ, run<0, 0, 0>(blocks, iters, src), run<1, 0, 0>(blocks, iters, src),
           run<1, 1, 0>(blocks, iters, src), run<1, 1, 1>(blocks, iters, src), run<1, 3, 1>(blocks, iters, src));
  }
  return 0;
}
