// Device helper shared by bn.hip and bn_multi.hip: the fixed-order fp64 sum of per-tile statistics partials.
#pragma once
#include "tbn_common.h"

// Sums partial[(i*2 + s) * pld + c] over i < nparts for the 16 channels [c0, c0 + 16) and s in {0, 1}.
// 256 threads = 4 float4 channel quads x 64 row slots: a thread walks rows slot, slot + 64, ... with four independent
// 16-B loads in flight per statistic (these kernels sit on the conv -> BN -> conv dependency chain and are pure
// latency: the former 8 channels x 32 slots / scalar-load form took 9 us for 588 rows, 78 us for the stem's 9408),
// then 32 threads add the 64 slot sums in fixed order -- the result does not depend on scheduling (deterministic).
// Returns the two sums of channel c0 + tid in threads tid < 16 (others: zeros); `red` = 64*32 doubles of LDS.
__device__ __forceinline__ void tbn_sum_partials16(const float* __restrict__ partial, int pld, int nparts, int c0, int C,
                                                   double* red, double* out_s1, double* out_s2) {
  const int tid = threadIdx.x, cq = tid & 3, slot = tid >> 2;
  const int c = c0 + cq * 4;
  double a0 = 0, a1 = 0, a2 = 0, a3 = 0, b0 = 0, b1 = 0, b2 = 0, b3 = 0;
  if (c < C) {
    const float* p = partial + c;
    int i = slot;
    for (; i + 192 < nparts; i += 256) {
      float4 u[4], v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        u[k] = *reinterpret_cast<const float4*>(p + ((size_t)(i + 64 * k) * 2 + 0) * pld);
        v[k] = *reinterpret_cast<const float4*>(p + ((size_t)(i + 64 * k) * 2 + 1) * pld);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        a0 += (double)u[k].x; a1 += (double)u[k].y; a2 += (double)u[k].z; a3 += (double)u[k].w;
        b0 += (double)v[k].x; b1 += (double)v[k].y; b2 += (double)v[k].z; b3 += (double)v[k].w;
      }
    }
    for (; i < nparts; i += 64) {
      const float4 u = *reinterpret_cast<const float4*>(p + ((size_t)i * 2 + 0) * pld);
      const float4 v = *reinterpret_cast<const float4*>(p + ((size_t)i * 2 + 1) * pld);
      a0 += (double)u.x; a1 += (double)u.y; a2 += (double)u.z; a3 += (double)u.w;
      b0 += (double)v.x; b1 += (double)v.y; b2 += (double)v.z; b3 += (double)v.w;
    }
  }
  double* r = red + slot * 32 + cq * 4;
  r[0] = a0; r[1] = a1; r[2] = a2; r[3] = a3;
  r[16] = b0; r[17] = b1; r[18] = b2; r[19] = b3;
  __syncthreads();
  double s = 0.0;
  if (tid < 32) {
#pragma unroll 8
    for (int k = 0; k < 64; ++k) s += red[k * 32 + tid];
  }
  // threads 0..15 hold S1 of their channel, 16..31 S2: hand S2 over through LDS
  __syncthreads();
  if (tid >= 16 && tid < 32) red[tid - 16] = s;
  __syncthreads();
  *out_s1 = (tid < 16) ? s : 0.0;
  *out_s2 = (tid < 16) ? red[tid] : 0.0;
}
