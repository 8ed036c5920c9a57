// Batched ("multi-tensor") training-BN kernels: the BN steps of up to four INDEPENDENT BN layers in one launch.
//
// Inside an inception block the 3x3, double_3x3_1 and pool_proj convs only depend on the fused 1x1 group, so
// their BN finalize / apply (forward) and BN-backward reduce / finalize / apply form three identical little
// kernel chains.  On this GPU every kernel on a stream's dependency chain costs ~9 us of step time no matter how
// short it runs (measured, DESIGN.md finding 6), and a third of the ~1000 BN launches of a step belong to such
// triples.  These kernels take up to three layer descriptors as kernel arguments; a workgroup finds its layer
// with a scalar scan of the block-offset table and then runs the single-layer code of bn.hip on it (same
// arithmetic, same summation order: results are bit-identical to the per-layer launches).
#include <cstdlib>
#include <cstring>
#include "tbn_common.h"
#include "tbn_kernels.h"
#include "tbn_bn_dev.h"
#include "tbn_rider_dev.h"

namespace {

__device__ __forceinline__ int find_layer(const int* blk0, int n, int b) {
  int l = 0;
  while (l + 1 < n && b >= blk0[l + 1]) ++l;
  return l;
}

// Timing diagnostics only (-DTBN_DIAG=1 build, never shipped; results are INVALID): TBN_DIAG_SKIP bit 16 drops the forward
// finalize launches of the batched BN steps, bit 32 the backward ones -- what these ~6-us launches cost the step, per pass
#ifndef TBN_DIAG
#define TBN_DIAG 0
#endif
inline bool diag_skip_fin(int bit) {
#if TBN_DIAG
  static const int mask = getenv("TBN_DIAG_SKIP") ? atoi(getenv("TBN_DIAG_SKIP")) : 0;
  return (mask & bit) != 0;
#else
  (void)bit;
  return false;
#endif
}

inline int ew_grid(size_t items) {
  size_t g = (items + 255) / 256;
  return (int)(g > 4096 ? 4096 : (g < 1 ? 1 : g));
}

}  // namespace

// ---------------------------------------------------------------- forward: finalize
__global__ __launch_bounds__(256) void bn_finalize_multi_kernel(BnFwdBatch b) {
  __shared__ double red[64 * 32];
  const int li = find_layer(b.fin_blk0, b.n, blockIdx.x);
  const BnFwdLayer& L = b.l[li];
  const int blk = blockIdx.x - b.fin_blk0[li];
  const int C = L.C, P = L.P;
  const int tid = threadIdx.x;
  const int c = blk * TBN_FIN_CH + tid;
  double s1, s2;
  tbn_sum_partials<TBN_FIN_CH>(L.partial, L.pld, L.nparts, blk * TBN_FIN_CH, C, red, &s1, &s2);
  if (tid < TBN_FIN_CH && c < C) {
    const double mean = s1 / P;
    double var = s2 / P - mean * mean;
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)b.eps));
    const float sc = L.gamma[c] * rstd;
    L.save_mean[c] = (float)mean;
    L.save_rstd[c] = rstd;
    L.scale[c] = sc;
    L.shift[c] = fmaf(-(float)mean, sc, L.beta[c]);
    if (L.running_mean != nullptr) {
      const double unb = P > 1 ? var * ((double)P / (double)(P - 1)) : var;
      const float mean_b = (float)mean + (L.conv_bias != nullptr ? L.conv_bias[c] : 0.f);
      L.running_mean[c] = (1.f - b.momentum) * L.running_mean[c] + b.momentum * mean_b;
      L.running_var[c] = (1.f - b.momentum) * L.running_var[c] + b.momentum * (float)unb;
    }
  }
}

// ---------------------------------------------------------------- forward: apply z = relu(y*scale+shift)
// A thread owns ONE float4 channel quad (tid % G) and walks rows tid / G, + RP, ... of its workgroup's row block: the
// per-channel scale / shift are loaded once per thread and the element index needs no division (the former flat
// grid-stride form spent ~20 VALU instructions per float4 on i / G and re-read scale / shift for every element --
// VALU these HBM-bound kernels take from the fp32 MFMAs running beside them).  Four rows in flight per thread.
__global__ __launch_bounds__(256) void bn_apply_multi_kernel(BnFwdBatch b) {
  const int li = find_layer(b.app_blk0, b.n, blockIdx.x);
  const BnFwdLayer& L = b.l[li];
  const int blk = blockIdx.x - b.app_blk0[li];
  const int p0 = blk * L.app_rows, p1 = min(L.P, p0 + L.app_rows);
  if (L.nseg == 1 && L.seg[0].col_begin == 0) {   // (every engine launch) the code the conv-launch riders run as well
    tbn_bn_apply_rows(L.y, L.y_ld, L.seg[0].ptr, L.seg[0].ld, L.scale, L.shift, L.C, p0, p1);
    return;
  }
  const int G = L.C >> 2, RP = 256 / G;
  const int cg = threadIdx.x % G, rs = threadIdx.x / G;
  if (rs >= RP) return;
  const int c = cg * 4;
  const float4 sc = *reinterpret_cast<const float4*>(L.scale + c);
  const float4 sh = *reinterpret_cast<const float4*>(L.shift + c);
  int sg = 0;
  if (L.nseg > 1 && c >= L.seg[1].col_begin) sg = 1;
  if (L.nseg > 2 && c >= L.seg[2].col_begin) sg = 2;
  const float* yp = L.y + (size_t)(p0 + rs) * L.y_ld + c;
  float* zp = L.seg[sg].ptr + (size_t)(p0 + rs) * L.seg[sg].ld + (c - L.seg[sg].col_begin);
  const size_t ystep = (size_t)RP * L.y_ld, zstep = (size_t)RP * L.seg[sg].ld;
  int p = p0 + rs;
  for (; p + 3 * RP < p1; p += 4 * RP) {
    float4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = *reinterpret_cast<const float4*>(yp + k * ystep);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float4 z;
      z.x = fmaxf(fmaf(v[k].x, sc.x, sh.x), 0.f);
      z.y = fmaxf(fmaf(v[k].y, sc.y, sh.y), 0.f);
      z.z = fmaxf(fmaf(v[k].z, sc.z, sh.z), 0.f);
      z.w = fmaxf(fmaf(v[k].w, sc.w, sh.w), 0.f);
      *reinterpret_cast<float4*>(zp + k * zstep) = z;
    }
    yp += 4 * ystep;
    zp += 4 * zstep;
  }
  for (; p < p1; p += RP) {
    const float4 v = *reinterpret_cast<const float4*>(yp);
    float4 z;
    z.x = fmaxf(fmaf(v.x, sc.x, sh.x), 0.f);
    z.y = fmaxf(fmaf(v.y, sc.y, sh.y), 0.f);
    z.z = fmaxf(fmaf(v.z, sc.z, sh.z), 0.f);
    z.w = fmaxf(fmaf(v.w, sc.w, sh.w), 0.f);
    *reinterpret_cast<float4*>(zp) = z;
    yp += ystep;
    zp += zstep;
  }
}

// rows per workgroup of the apply kernels: 4 .. 16 passes of the row lanes, aiming at ~2048 workgroups per layer
static inline int apply_rows(int P, int C) {
  const int RP = 256 / (C / 4);
  int it = cdiv(P, RP * 2048);
  if (it < 4) it = 4;
  if (it > 16) it = 16;
  return RP * it;
}

int tbn_launch_bn_fwd_multi(BnFwdBatch& b, hipStream_t st) { return tbn_launch_bn_fwd_multi_defer(b, 0u, nullptr, st); }

int tbn_rider_place(RiderP* r, int gemm_blocks) {
  // behind the GEMM's workgroups by default: the dispatcher hands out workgroups in grid order, the GEMM tiles are the
  // long jobs and the rider's short HBM-bound workgroups fill the CUs the last round of tiles leaves idle (longest first).
  // TBN_RIDER_FRONT=1 (A/B knob) puts them in front instead, padded to a multiple of 8 so that the XCD-aware tile maps hold.
  static const int front = tbn_env_int("TBN_RIDER_FRONT", 0, 0, 1);
  if (r == nullptr || r->nblk <= 0) return gemm_blocks;
  if (front) {
    r->span = (r->nblk + 7) / 8 * 8;
    r->first = 0;
    r->gemm0 = r->span;
  } else {
    r->span = r->nblk;
    r->first = gemm_blocks;
    r->gemm0 = 0;
  }
  return gemm_blocks + r->span;
}

int tbn_launch_bn_fwd_multi_defer(BnFwdBatch& b, unsigned defer_mask, RiderP* rider, hipStream_t st) {
  TBN_REQUIRE(b.n >= 1 && b.n <= TBN_BN_MAXL, "bn_fwd_multi: %d layers", b.n);
  TBN_REQUIRE(defer_mask == 0u || rider != nullptr, "bn_fwd_multi: deferred members need a rider descriptor");
  b.fin_blk0[0] = b.app_blk0[0] = 0;
  for (int i = 0; i < b.n; ++i) {
    BnFwdLayer& L = b.l[i];
    TBN_REQUIRE(L.C % 4 == 0 && L.C <= 1024 && L.y_ld % 4 == 0 && L.y_ld >= L.C && L.pld >= L.C && L.nseg >= 1 && L.nseg <= 3,
                "bn_fwd_multi: bad C / pitch / nseg");
    for (int s = 0; s < L.nseg; ++s)
      TBN_REQUIRE(L.seg[s].ld % 4 == 0 && L.seg[s].col_begin % 4 == 0, "bn_fwd_multi: segment pitch/offset must be x4");
    b.fin_blk0[i + 1] = b.fin_blk0[i] + cdiv(L.C, TBN_FIN_CH);
    L.app_rows = apply_rows(L.P, L.C);
    b.app_blk0[i + 1] = b.app_blk0[i] + cdiv(L.P, L.app_rows);
  }
  if (!diag_skip_fin(16)) TBN_KLAUNCH(bn_finalize_multi_kernel, dim3(b.fin_blk0[b.n]), dim3(256), 0, st, b);
  TBN_CHECK_LAUNCH("bn_finalize_multi");
  if (defer_mask == 0u) {
    TBN_KLAUNCH(bn_apply_multi_kernel, dim3(b.app_blk0[b.n]), dim3(256), 0, st, b);
    TBN_CHECK_LAUNCH("bn_apply_multi");
    return TBN_OK;
  }
  // members whose apply pass rides in a later conv launch: described in `rider`; the others run here, compacted
  static thread_local BnFwdBatch now;
  now.n = 0;
  now.momentum = b.momentum;
  now.eps = b.eps;
  now.app_blk0[0] = 0;
  memset(rider, 0, sizeof(*rider));
  rider->kind = 1;
  for (int i = 0; i < b.n; ++i) {
    const BnFwdLayer& L = b.l[i];
    const int blocks = b.app_blk0[i + 1] - b.app_blk0[i];
    if (defer_mask & (1u << i)) {
      TBN_REQUIRE(rider->n < TBN_RIDER_MAXL && L.nseg == 1 && L.seg[0].col_begin == 0, "bn_fwd_multi: rider holds at most %d single-segment layers", TBN_RIDER_MAXL);
      RiderLayer& R = rider->l[rider->n];
      R.y = L.y; R.y_ld = L.y_ld; R.out = L.seg[0].ptr; R.out_ld = L.seg[0].ld;
      R.scale = L.scale; R.shift = L.shift; R.P = L.P; R.C = L.C; R.rows = L.app_rows;
      rider->blk0[rider->n + 1] = rider->blk0[rider->n] + blocks;
      ++rider->n;
    } else {
      now.l[now.n] = L;
      now.app_blk0[now.n + 1] = now.app_blk0[now.n] + blocks;
      ++now.n;
    }
  }
  rider->nblk = rider->blk0[rider->n];
  if (now.n > 0) {
    TBN_KLAUNCH(bn_apply_multi_kernel, dim3(now.app_blk0[now.n]), dim3(256), 0, st, now);
    TBN_CHECK_LAUNCH("bn_apply_multi");
  }
  return TBN_OK;
}

// ---------------------------------------------------------------- backward: reduce
// g = dz * [y*scale+shift > 0];  xhat = (y-mean)*rstd;  S1 = sum g, S2 = sum g*xhat   (per-workgroup partials)
__global__ __launch_bounds__(256) void bn_bwd_reduce_multi_kernel(BnBwdBatch b) {
  __shared__ float red[2 * 2048];
  const int li = find_layer(b.red_blk0, b.n, blockIdx.x);
  const BnBwdLayer& L = b.l[li];
  const int blk = blockIdx.x - b.red_blk0[li];
  const int C = L.C, P = L.P, pch = L.pch;
  const int G = C >> 2, RP = 256 / G;
  const int tid = threadIdx.x, cg = tid % G, rs = tid / G, c = cg * 4;
  const int p0 = blk * pch, p1 = min(P, p0 + pch);
  float4 s1 = make_float4(0, 0, 0, 0), s2 = s1;
  if (rs < RP) {
    int sg = 0;
    if (L.nseg > 1 && c >= L.dz[1].col_begin) sg = 1;
    if (L.nseg > 2 && c >= L.dz[2].col_begin) sg = 2;
    const float* dzp = L.dz[sg].ptr + (c - L.dz[sg].col_begin);
    const int dld = L.dz[sg].ld;
    const float4 sc = *reinterpret_cast<const float4*>(L.scale + c);
    const float4 sh = *reinterpret_cast<const float4*>(L.shift + c);
    const float4 mu = *reinterpret_cast<const float4*>(L.mean + c);
    const float4 rs4 = *reinterpret_cast<const float4*>(L.rstd + c);
    const float* y = L.y;
    const int yld = L.y_ld;
    for (int p = p0 + rs; p < p1; p += RP) {
      const float4 v = *reinterpret_cast<const float4*>(y + (size_t)p * yld + c);
      const float4 d = *reinterpret_cast<const float4*>(dzp + (size_t)p * dld);
      const float gx = fmaf(v.x, sc.x, sh.x) > 0.f ? d.x : 0.f;
      const float gy = fmaf(v.y, sc.y, sh.y) > 0.f ? d.y : 0.f;
      const float gz = fmaf(v.z, sc.z, sh.z) > 0.f ? d.z : 0.f;
      const float gw = fmaf(v.w, sc.w, sh.w) > 0.f ? d.w : 0.f;
      s1.x += gx; s1.y += gy; s1.z += gz; s1.w += gw;
      s2.x = fmaf(gx, (v.x - mu.x) * rs4.x, s2.x);
      s2.y = fmaf(gy, (v.y - mu.y) * rs4.y, s2.y);
      s2.z = fmaf(gz, (v.z - mu.z) * rs4.z, s2.z);
      s2.w = fmaf(gw, (v.w - mu.w) * rs4.w, s2.w);
    }
    *reinterpret_cast<float4*>(&red[(rs * G + cg) * 4]) = s1;
    *reinterpret_cast<float4*>(&red[2048 + (rs * G + cg) * 4]) = s2;
  }
  __syncthreads();
  for (int cc = tid; cc < C; cc += 256) {
    float a = 0.f, s = 0.f;
    for (int r = 0; r < RP; ++r) {
      a += red[r * C + cc];
      s += red[2048 + r * C + cc];
    }
    L.partial[((size_t)blk * 2 + 0) * C + cc] = a;
    L.partial[((size_t)blk * 2 + 1) * C + cc] = s;
  }
}

// coef[0][c]=a, coef[1][c]=b, coef[2][c]=cst with dy = a*g + b*y + cst
__global__ __launch_bounds__(256) void bn_bwd_finalize_multi_kernel(BnBwdBatch b) {
  __shared__ double red[64 * 32];
  const int li = find_layer(b.fin_blk0, b.n, blockIdx.x);
  const BnBwdLayer& L = b.l[li];
  const int blk = blockIdx.x - b.fin_blk0[li];
  const int C = L.C, P = L.P;
  const int tid = threadIdx.x;
  const int c = blk * TBN_FIN_CH + tid;
  double s1, s2;
  tbn_sum_partials<TBN_FIN_CH>(L.partial, C, L.nparts, blk * TBN_FIN_CH, C, red, &s1, &s2);
  if (tid < TBN_FIN_CH && c < C) {
    const double sc = L.scale[c], rs = L.rstd[c], mu = L.mean[c];
    const double bb = -sc * rs * (s2 / P);
    L.coef[c] = (float)sc;
    L.coef[C + c] = (float)bb;
    L.coef[2 * C + c] = (float)(-sc * (s1 / P) - bb * mu);
    if (L.dgamma) L.dgamma[c] = (float)s2;
    if (L.dbeta) L.dbeta[c] = (float)s1;
    if (L.dbias) L.dbias[c] = 0.f;   // a per-channel constant added before a batch-stat BN has exactly zero gradient
  }
}

// y and dy alias (the engine converts y to dy in place).  Same thread mapping as bn_apply_multi_kernel: per-channel
// coefficients loaded once per thread, no index division, four rows in flight.
__global__ __launch_bounds__(256) void bn_bwd_apply_multi_kernel(BnBwdBatch b) {
  const int li = find_layer(b.app_blk0, b.n, blockIdx.x);
  const BnBwdLayer& L = b.l[li];
  const int blk = blockIdx.x - b.app_blk0[li];
  const int p0 = blk * L.app_rows, p1 = min(L.P, p0 + L.app_rows);
  if (L.nseg == 1 && L.dz[0].col_begin == 0) {   // (every engine launch) the code the conv-launch riders run as well
    tbn_bn_bwd_apply_rows(L.dz[0].ptr, L.dz[0].ld, L.y, L.y_ld, L.dy, L.scale, L.shift, L.coef, L.C, p0, p1);
    return;
  }
  const int C = L.C, G = C >> 2, RP = 256 / G;
  const int cg = threadIdx.x % G, rs = threadIdx.x / G;
  if (rs >= RP) return;
  const int c = cg * 4;
  int sg = 0;
  if (L.nseg > 1 && c >= L.dz[1].col_begin) sg = 1;
  if (L.nseg > 2 && c >= L.dz[2].col_begin) sg = 2;
  const float4 sc = *reinterpret_cast<const float4*>(L.scale + c);
  const float4 sh = *reinterpret_cast<const float4*>(L.shift + c);
  const float4 ca = *reinterpret_cast<const float4*>(L.coef + c);
  const float4 cb = *reinterpret_cast<const float4*>(L.coef + C + c);
  const float4 cc = *reinterpret_cast<const float4*>(L.coef + 2 * C + c);
  const float* dp = L.dz[sg].ptr + (size_t)(p0 + rs) * L.dz[sg].ld + (c - L.dz[sg].col_begin);
  const float* yp = L.y + (size_t)(p0 + rs) * L.y_ld + c;
  float* op = L.dy + (size_t)(p0 + rs) * L.y_ld + c;
  const size_t dstep = (size_t)RP * L.dz[sg].ld, ystep = (size_t)RP * L.y_ld;
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
      d[k] = *reinterpret_cast<const float4*>(dp + k * dstep);
      v[k] = *reinterpret_cast<const float4*>(yp + k * ystep);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) *reinterpret_cast<float4*>(op + k * ystep) = one(d[k], v[k]);
    dp += 4 * dstep;
    yp += 4 * ystep;
    op += 4 * ystep;
  }
  for (; p < p1; p += RP) {
    *reinterpret_cast<float4*>(op) = one(*reinterpret_cast<const float4*>(dp), *reinterpret_cast<const float4*>(yp));
    dp += dstep;
    yp += ystep;
    op += ystep;
  }
}

int tbn_launch_bn_bwd_multi(BnBwdBatch& b, hipStream_t st) { return tbn_launch_bn_bwd_multi_defer(b, 0u, nullptr, st); }

int tbn_launch_bn_bwd_multi_defer(BnBwdBatch& b, unsigned defer_mask, RiderP* rider, hipStream_t st) {
  TBN_REQUIRE(b.n >= 1 && b.n <= TBN_BN_MAXL, "bn_bwd_multi: %d layers", b.n);
  TBN_REQUIRE(defer_mask == 0u || rider != nullptr, "bn_bwd_multi: deferred members need a rider descriptor");
  b.red_blk0[0] = b.fin_blk0[0] = b.app_blk0[0] = 0;
  for (int i = 0; i < b.n; ++i) {
    BnBwdLayer& L = b.l[i];
    TBN_REQUIRE(L.C % 4 == 0 && L.C <= 1024 && L.y_ld % 4 == 0 && L.y_ld >= L.C && L.nseg >= 1 && L.nseg <= 3,
                "bn_bwd_multi: bad C / pitch / nseg");
    for (int s = 0; s < L.nseg; ++s)
      TBN_REQUIRE(L.dz[s].ld % 4 == 0 && L.dz[s].col_begin % 4 == 0, "bn_bwd_multi: segment pitch/offset must be x4");
    {  // same chunking as bn.hip's single-layer launch: rows per workgroup rounded to the row-lane count
      const int rp = 256 / (L.C / 4);
      int pch = cdiv(L.P, 512);
      if (pch < 64) pch = 64;
      L.pch = cdiv(pch, rp) * rp;
      L.nparts = cdiv(L.P, L.pch);
    }
    int red_blocks = L.nparts;
    if (L.ext_parts > 0) {   // partials came from the data-gradient epilogue that finished dz: nothing to reduce
      L.nparts = L.ext_parts;
      red_blocks = 0;
    }
    b.red_blk0[i + 1] = b.red_blk0[i] + red_blocks;
    b.fin_blk0[i + 1] = b.fin_blk0[i] + cdiv(L.C, TBN_FIN_CH);
    L.app_rows = apply_rows(L.P, L.C);
    b.app_blk0[i + 1] = b.app_blk0[i] + cdiv(L.P, L.app_rows);
  }
  if (b.red_blk0[b.n] > 0) {
    TBN_KLAUNCH(bn_bwd_reduce_multi_kernel, dim3(b.red_blk0[b.n]), dim3(256), 0, st, b);
    TBN_CHECK_LAUNCH("bn_bwd_reduce_multi");
  }
  if (!diag_skip_fin(32)) TBN_KLAUNCH(bn_bwd_finalize_multi_kernel, dim3(b.fin_blk0[b.n]), dim3(256), 0, st, b);
  TBN_CHECK_LAUNCH("bn_bwd_finalize_multi");
  if (defer_mask == 0u) {
    TBN_KLAUNCH(bn_bwd_apply_multi_kernel, dim3(b.app_blk0[b.n]), dim3(256), 0, st, b);
    TBN_CHECK_LAUNCH("bn_bwd_apply_multi");
    return TBN_OK;
  }
  static thread_local BnBwdBatch now;
  now.n = 0;
  now.app_blk0[0] = 0;
  memset(rider, 0, sizeof(*rider));
  rider->kind = 2;
  for (int i = 0; i < b.n; ++i) {
    const BnBwdLayer& L = b.l[i];
    const int blocks = b.app_blk0[i + 1] - b.app_blk0[i];
    if (defer_mask & (1u << i)) {
      TBN_REQUIRE(rider->n < TBN_RIDER_MAXL && L.nseg == 1 && L.dz[0].col_begin == 0, "bn_bwd_multi: rider holds at most %d single-segment layers", TBN_RIDER_MAXL);
      RiderLayer& R = rider->l[rider->n];
      R.y = L.y; R.y_ld = L.y_ld; R.out = L.dy; R.out_ld = L.y_ld; R.dz = L.dz[0].ptr; R.dz_ld = L.dz[0].ld;
      R.scale = L.scale; R.shift = L.shift; R.coef = L.coef; R.P = L.P; R.C = L.C; R.rows = L.app_rows;
      rider->blk0[rider->n + 1] = rider->blk0[rider->n] + blocks;
      ++rider->n;
    } else {
      now.l[now.n] = L;
      now.app_blk0[now.n + 1] = now.app_blk0[now.n] + blocks;
      ++now.n;
    }
  }
  rider->nblk = rider->blk0[rider->n];
  if (now.n > 0) {
    TBN_KLAUNCH(bn_bwd_apply_multi_kernel, dim3(now.app_blk0[now.n]), dim3(256), 0, st, now);
    TBN_CHECK_LAUNCH("bn_bwd_apply_multi");
  }
  return TBN_OK;
}
