// Device helper shared by bn.hip and bn_multi.hip: the fixed-order fp64 sum of per-tile statistics partials.
#pragma once
#include "tbn_common.h"

// Channels per finalize workgroup (a build-time knob: -DTBN_FIN_CH=4 | 8 | 16).  The finalize kernels sit on the
// conv -> BN -> conv dependency chain and are pure latency (a step's ~190 of them cost 1.16 ms of step time at ~6 us
// each, DESIGN.md finding 14): round 2 went from 8 channels x 32 row slots with scalar loads to 16 channels x 64 slots
// with 16-B loads (9 -> 3 us for 588 rows, -0.97 ms per step).  Round 4 measured the next step of the same idea -- ONE
// channel quad per workgroup and 256 row slots: 4x the workgroups, one round of loads instead of three for 588 rows --
// and it bought nothing (same-box A/B, three alternations: 36.05 vs 35.98 ms; config 2 12.31 vs 12.23; config 3 50.53 vs
// 50.42; profiles/r04_ab_finalize_channels.txt): what a finalize costs now is its launch boundary on the chain, not the
// rows a thread walks.  16 stays.
#ifndef TBN_FIN_CH
#define TBN_FIN_CH 16
#endif

// Sums partial[(i*2 + s) * pld + c] over i < nparts for the CH channels [c0, c0 + CH) and s in {0, 1}.
// 256 threads = CH/4 float4 channel quads x S = 1024/CH row slots: a thread walks rows slot, slot + S, ... with four
// independent 16-B loads in flight per statistic, then the S slot sums are added in a fixed two-level order (groups of 8
// slots, then the groups): the result does not depend on scheduling (deterministic).
// Returns the two sums of channel c0 + tid in threads tid < CH (others: zeros); `red` = 2048 doubles of LDS.
template <int CH>
__device__ __forceinline__ void tbn_sum_partials(const float* __restrict__ partial, int pld, int nparts, int c0, int C,
                                                 double* red, double* out_s1, double* out_s2) {
  constexpr int Q = CH / 4, S = 256 / Q, V = Q * 8, G = S / 8;
  static_assert(CH == 4 || CH == 8 || CH == 16, "finalize: 4, 8 or 16 channels per workgroup");
  const int tid = threadIdx.x, cq = tid % Q, slot = tid / Q;
  const int c = c0 + cq * 4;
  double a0 = 0, a1 = 0, a2 = 0, a3 = 0, b0 = 0, b1 = 0, b2 = 0, b3 = 0;
  if (c < C) {
    const float* p = partial + c;
    for (int i = slot; i < nparts; i += 4 * S) {
      float4 u[4], v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int r = i + S * k;
        const bool ok = r < nparts;
        const size_t rr = (size_t)(ok ? r : i);          // an in-range row: the load is issued either way
        u[k] = *reinterpret_cast<const float4*>(p + (rr * 2 + 0) * pld);
        v[k] = *reinterpret_cast<const float4*>(p + (rr * 2 + 1) * pld);
        if (!ok) u[k] = v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        a0 += (double)u[k].x; a1 += (double)u[k].y; a2 += (double)u[k].z; a3 += (double)u[k].w;
        b0 += (double)v[k].x; b1 += (double)v[k].y; b2 += (double)v[k].z; b3 += (double)v[k].w;
      }
    }
  }
  double* r = red + slot * V + cq * 8;                   // per slot: per quad a0..a3 b0..b3
  r[0] = a0; r[1] = a1; r[2] = a2; r[3] = a3;
  r[4] = b0; r[5] = b1; r[6] = b2; r[7] = b3;
  __syncthreads();
  const int vi = tid % V, g = tid / V;                   // level 1: value vi of slot group g (8 slots), V * G = 256 threads
  double s = 0.0;
#pragma unroll
  for (int k = 0; k < 8; ++k) s += red[(g * 8 + k) * V + vi];
  __syncthreads();
  red[g * V + vi] = s;
  __syncthreads();
  double t = 0.0;
  if (tid < V) {                                         // level 2: the G group sums of value tid
#pragma unroll 8
    for (int gg = 0; gg < G; ++gg) t += red[gg * V + tid];
  }
  __syncthreads();
  if (tid < V) red[tid] = t;
  __syncthreads();
  // channel j of the workgroup = quad j / 4, component j % 4: S1 at value (j/4)*8 + j%4, S2 four further
  *out_s1 = (tid < CH) ? red[(tid >> 2) * 8 + (tid & 3)] : 0.0;
  *out_s2 = (tid < CH) ? red[(tid >> 2) * 8 + 4 + (tid & 3)] : 0.0;
}
