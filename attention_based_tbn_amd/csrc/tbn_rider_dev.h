// Device code of the elementwise training-BN passes, shared by the stand-alone batched kernels (bn_multi.hip) and the
// rider workgroups of the conv-GEMM launches (conv_igemm.hip): one definition, so a pass gives the same bits wherever
// it runs.  Thread mapping (VALU diet, round 2): a thread owns ONE float4 channel quad (tid % G) and walks the rows
// tid / G, + RP, ... of its workgroup's row block -- coefficients loaded once, no index division, four rows in flight.
#pragma once
#include "tbn_common.h"
#include "tbn_kernels.h"

// z = relu(y * scale + shift) for rows [p0, p1) of a (P, C) column range; z_col0 = channel offset of `z` already applied
__device__ __forceinline__ void tbn_bn_apply_rows(const float* __restrict__ y, int y_ld, float* __restrict__ z, int z_ld,
                                                  const float* __restrict__ scale, const float* __restrict__ shift, int C,
                                                  int p0, int p1) {
  const int G = C >> 2, RP = 256 / G;
  const int cg = threadIdx.x % G, rs = threadIdx.x / G;
  if (rs >= RP) return;
  const int c = cg * 4;
  const float4 sc = *reinterpret_cast<const float4*>(scale + c);
  const float4 sh = *reinterpret_cast<const float4*>(shift + c);
  const float* yp = y + (size_t)(p0 + rs) * y_ld + c;
  float* zp = z + (size_t)(p0 + rs) * z_ld + c;
  const size_t ystep = (size_t)RP * y_ld, zstep = (size_t)RP * z_ld;
  int p = p0 + rs;
  for (; p + 3 * RP < p1; p += 4 * RP) {
    float4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = tbn_ld4<(TBN_BN_NT & 1) != 0>(yp + k * ystep);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float4 o;
      o.x = fmaxf(fmaf(v[k].x, sc.x, sh.x), 0.f);
      o.y = fmaxf(fmaf(v[k].y, sc.y, sh.y), 0.f);
      o.z = fmaxf(fmaf(v[k].z, sc.z, sh.z), 0.f);
      o.w = fmaxf(fmaf(v[k].w, sc.w, sh.w), 0.f);
      tbn_st4<(TBN_BN_NT & 4) != 0>(zp + k * zstep, o);
    }
    yp += 4 * ystep;
    zp += 4 * zstep;
  }
  for (; p < p1; p += RP) {
    const float4 v = tbn_ld4<(TBN_BN_NT & 1) != 0>(yp);
    float4 o;
    o.x = fmaxf(fmaf(v.x, sc.x, sh.x), 0.f);
    o.y = fmaxf(fmaf(v.y, sc.y, sh.y), 0.f);
    o.z = fmaxf(fmaf(v.z, sc.z, sh.z), 0.f);
    o.w = fmaxf(fmaf(v.w, sc.w, sh.w), 0.f);
    tbn_st4<(TBN_BN_NT & 4) != 0>(zp, o);
    yp += ystep;
    zp += zstep;
  }
}

// dy = a * [y*scale+shift > 0] dz + b * y + c  (coef = a | b | c rows of C floats); dy may alias y
__device__ __forceinline__ void tbn_bn_bwd_apply_rows(const float* __restrict__ dz, int dz_ld, const float* y, int y_ld,
                                                      float* dy, const float* __restrict__ scale,
                                                      const float* __restrict__ shift, const float* __restrict__ coef, int C,
                                                      int p0, int p1) {
  const int G = C >> 2, RP = 256 / G;
  const int cg = threadIdx.x % G, rs = threadIdx.x / G;
  if (rs >= RP) return;
  const int c = cg * 4;
  const float4 sc = *reinterpret_cast<const float4*>(scale + c);
  const float4 sh = *reinterpret_cast<const float4*>(shift + c);
  const float4 ca = *reinterpret_cast<const float4*>(coef + c);
  const float4 cb = *reinterpret_cast<const float4*>(coef + C + c);
  const float4 cc = *reinterpret_cast<const float4*>(coef + 2 * C + c);
  const float* dp = dz + (size_t)(p0 + rs) * dz_ld + c;
  const float* yp = y + (size_t)(p0 + rs) * y_ld + c;
  float* op = dy + (size_t)(p0 + rs) * y_ld + c;
  const size_t dstep = (size_t)RP * dz_ld, ystep = (size_t)RP * y_ld;
  auto one = [&](const float4 d, const float4 v) {
    float4 o;
    o.x = fmaf(ca.x, fmaf(v.x, sc.x, sh.x) > 0.f ? d.x : 0.f, fmaf(cb.x, v.x, cc.x));
    o.y = fmaf(ca.y, fmaf(v.y, sc.y, sh.y) > 0.f ? d.y : 0.f, fmaf(cb.y, v.y, cc.y));
    o.z = fmaf(ca.z, fmaf(v.z, sc.z, sh.z) > 0.f ? d.z : 0.f, fmaf(cb.z, v.z, cc.z));
    o.w = fmaf(ca.w, fmaf(v.w, sc.w, sh.w) > 0.f ? d.w : 0.f, fmaf(cb.w, v.w, cc.w));
    return o;
  };
  int p = p0 + rs;
  for (; p + 3 * RP < p1; p += 4 * RP) {
    float4 d[4], v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      d[k] = tbn_ld4<(TBN_BN_NT & 2) != 0>(dp + k * dstep);
      v[k] = tbn_ld4<(TBN_BN_NT & 2) != 0>(yp + k * ystep);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) tbn_st4<(TBN_BN_NT & 4) != 0>(op + k * ystep, one(d[k], v[k]));
    dp += 4 * dstep;
    yp += 4 * ystep;
    op += 4 * ystep;
  }
  for (; p < p1; p += RP) {
    tbn_st4<(TBN_BN_NT & 4) != 0>(op, one(tbn_ld4<(TBN_BN_NT & 2) != 0>(dp), tbn_ld4<(TBN_BN_NT & 2) != 0>(yp)));
    dp += dstep;
    yp += ystep;
    op += ystep;
  }
}

// one rider workgroup (rb = its index inside the rider's part of the grid)
__device__ __forceinline__ void tbn_rider_block(const RiderP& r, int rb) {
  int li = 0;
  while (li + 1 < r.n && rb >= r.blk0[li + 1]) ++li;
  const RiderLayer& L = r.l[li];
  const int blk = rb - r.blk0[li];
  const int p0 = blk * L.rows, p1 = min(L.P, p0 + L.rows);
  if (r.kind == 1)
    tbn_bn_apply_rows(L.y, L.y_ld, L.out, L.out_ld, L.scale, L.shift, L.C, p0, p1);
  else
    tbn_bn_bwd_apply_rows(L.dz, L.dz_ld, L.y, L.y_ld, L.out, L.scale, L.shift, L.coef, L.C, p0, p1);
}

// first statement of a conv-GEMM kernel that accepts riders: rider workgroups do their pass and leave, the others
// continue as GEMM workgroup `bid`
#define TBN_RIDER_DISPATCH(r, bid)                                   \
  const int rider_rb__ = (int)blockIdx.x - (r).first;               \
  if (rider_rb__ >= 0 && rider_rb__ < (r).span) {                   \
    if (rider_rb__ < (r).nblk) tbn_rider_block((r), rider_rb__);    \
    return;                                                          \
  }                                                                  \
  const int bid = (int)blockIdx.x - (r).gemm0;
