// Does the <1,1> tile's single accumulator chain cost MFMA issue slots?  (round 6)
// The K-step of a <1,1> wave tile is 16 MFMAs on ONE accumulator, in four groups of 4 with the ds_read of the fragments two
// groups ahead issued between the groups (conv_igemm.hip: sched_group_barrier chain).  HISTORY finding 2 measured in
// isolation: a dependent MFMA is free directly behind its producer or >= 4 MFMAs later, but an LDS / VALU instruction
// between two dependent MFMAs costs +40 ... 80 % of an MFMA.  This loop reproduces the K-step's instruction order with
//   NACC = 1: every MFMA on the same accumulator (the shipped <1,1> order)
//   NACC = 4: MFMA q of a group on accumulator q (each accumulator is touched every 4th MFMA, whatever sits between)
// at 1 ... 4 waves per SIMD (other waves' MFMAs interleave on the pipe), fragments really read from LDS.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC, int READS>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[4 * 2 * 64 * 36];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* my = lds + wave * 2 * 64 * 36;
  for (int i = lane; i < 2 * 64 * 36; i += 64) my[i] = 1e-3f * (float)(i & 31);
  __syncthreads();
  f32x16 acc[NACC];
  for (int a = 0; a < NACC; ++a)
    for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
  const float4* A = reinterpret_cast<const float4*>(my + (lane & 31) * 36 + (lane >> 5) * 4);
  const float4* B = reinterpret_cast<const float4*>(my + 64 * 36 + (lane & 31) * 36 + (lane >> 5) * 4);
  float4 fa[2], fb[2];
  fa[0] = A[0]; fb[0] = B[0];
  fa[1] = A[2]; fb[1] = B[2];
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int buf = g & 1;
      acc[0 % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[buf].x, fb[buf].x, acc[0 % NACC], 0, 0, 0);
      acc[1 % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[buf].y, fb[buf].y, acc[1 % NACC], 0, 0, 0);
      acc[2 % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[buf].z, fb[buf].z, acc[2 % NACC], 0, 0, 0);
      acc[3 % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[buf].w, fb[buf].w, acc[3 % NACC], 0, 0, 0);
      if (READS) {   // fragments of the group after next into the buffer just consumed
        fa[buf] = A[2 * ((g + 2) & 3)];
        fb[buf] = B[2 * ((g + 2) & 3)];
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, READS ? 2 : 0, 0);
    }
  }
  float r = 0;
  for (int a = 0; a < NACC; ++a) r += acc[a][lane & 15];
  out[blockIdx.x * blockDim.x + tid] = r;
}
template <int NACC, int READS> double run(int blocks, int iters) {
  float* d; hipMalloc(&d, (size_t)blocks * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) k<NACC, READS><<<blocks, 256>>>(d, iters);     // warm the clock (finding 13)
  hipEventRecord(e0); k<NACC, READS><<<blocks, 256>>>(d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); hipFree(d);
  return 2.0 * 32 * 32 * 2 * 16.0 * iters * blocks * 4 / (ms * 1e-3) / 1e12;
}
int main() {
  const int iters = 4000;
  printf("TFLOP/s of the K-step's MFMA order (peak 157.3): waves/SIMD | 1 acc, no reads | 4 acc, no reads | 1 acc + reads | 4 acc + reads\n");
  for (int occ = 1; occ <= 4; ++occ) {
    const int blocks = 256 * occ;
    printf("  %d   %7.1f %7.1f %7.1f %7.1f\n", occ, run<1, 0>(blocks, iters), run<4, 0>(blocks, iters), run<1, 1>(blocks, iters),
           run<4, 1>(blocks, iters));
  }
  return 0;
}
