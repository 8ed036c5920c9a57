// BatchNorm2d (+ReLU) for NHWC fp32 activations on gfx950: HBM-bound streaming kernels.
//
// Reference semantics: every conv of the backbone is followed by nn.BatchNorm2d(C, affine=True)
// (eps 1e-5, momentum 0.1) + ReLU(inplace) (reference core/models/bn_inception_audio.py:24-31 ...).
// Training keeps the modules in .train() (core/tools/train.py:68) so *batch statistics* are used
// and running stats are updated even for the "partialbn"-frozen layers (core/models/model.py:164-176
// only freezes the affine parameters); eval (train.py:146, test.py:67) uses the running stats.
//
// Channel is the fastest dim, so a wave reads 16 B / lane fully coalesced and every lane owns 4
// fixed channels: per-channel reductions are register accumulations + one fixed-order LDS pass,
// written as per-workgroup partials and summed in fp64 by a finalize kernel (deterministic).
#include "tbn_common.h"
#include "tbn_kernels.h"

static inline int pick_chunk(int P, int rp) {
  int pch = cdiv(P, 512);
  if (pch < 64) pch = 64;
  return cdiv(pch, rp) * rp;
}

// ---------------------------------------------------------------- forward statistics (standalone)
__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ y, int ld, int P, int C, int pch,
                                                       float* __restrict__ partial) {
  __shared__ float red[2 * 2048];
  const int G = C >> 2, RP = 256 / G;
  const int tid = threadIdx.x, cg = tid % G, rs = tid / G;
  const int p0 = blockIdx.x * pch, p1 = min(P, p0 + pch);
  float4 s1 = make_float4(0, 0, 0, 0), s2 = s1;
  if (rs < RP)
    for (int p = p0 + rs; p < p1; p += RP) {
      const float4 v = *reinterpret_cast<const float4*>(y + (size_t)p * ld + cg * 4);
      s1.x += v.x; s1.y += v.y; s1.z += v.z; s1.w += v.w;
      s2.x = fmaf(v.x, v.x, s2.x); s2.y = fmaf(v.y, v.y, s2.y); s2.z = fmaf(v.z, v.z, s2.z); s2.w = fmaf(v.w, v.w, s2.w);
    }
  if (rs < RP) {
    *reinterpret_cast<float4*>(&red[(rs * G + cg) * 4]) = s1;
    *reinterpret_cast<float4*>(&red[2048 + (rs * G + cg) * 4]) = s2;
  }
  __syncthreads();
  for (int c = tid; c < C; c += 256) {
    float a = 0.f, b = 0.f;
    for (int r = 0; r < RP; ++r) {
      a += red[r * C + c];
      b += red[2048 + r * C + c];
    }
    partial[((size_t)blockIdx.x * 2 + 0) * C + c] = a;
    partial[((size_t)blockIdx.x * 2 + 1) * C + c] = b;
  }
}

int tbn_bn_stats_parts(int P, int C) {
  const int rp = 256 / (C / 4);
  return cdiv(P, pick_chunk(P, rp));
}

int tbn_launch_bn_stats(const float* y, int ld, int P, int C, float* partial, int* nparts, hipStream_t st) {
  TBN_REQUIRE(C % 4 == 0 && C <= 1024 && ld % 4 == 0, "bn_stats: C must be a multiple of 4 and <= 1024");
  const int rp = 256 / (C / 4), pch = pick_chunk(P, rp), parts = cdiv(P, pch);
  hipLaunchKernelGGL(bn_stats_kernel, dim3(parts), dim3(256), 0, st, y, ld, P, C, pch, partial);
  TBN_CHECK_LAUNCH("bn_stats");
  if (nparts) *nparts = parts;
  return TBN_OK;
}

// ---------------------------------------------------------------- finalize: partials -> mean/rstd/scale/shift
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ partial, int nparts, int P, int C,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ beta,
                                                          const float* __restrict__ conv_bias, float* running_mean,
                                                          float* running_var, float momentum, float eps,
                                                          float* save_mean, float* save_rstd, float* scale,
                                                          float* shift) {
  // 8 channels per workgroup, 32 slots per channel: short dependent chains, fixed summation order
  __shared__ double red[2][32][8];
  const int tid = threadIdx.x, cl = tid & 7, slot = tid >> 3;
  const int c = blockIdx.x * 8 + cl;
  double a = 0.0, b = 0.0;
  if (c < C)
    for (int i = slot; i < nparts; i += 32) {
      a += (double)partial[((size_t)i * 2 + 0) * C + c];
      b += (double)partial[((size_t)i * 2 + 1) * C + c];
    }
  red[0][slot][cl] = a;
  red[1][slot][cl] = b;
  __syncthreads();
  if (slot == 0 && c < C) {
    double s1 = 0.0, s2 = 0.0;
    for (int k = 0; k < 32; ++k) {
      s1 += red[0][k][cl];
      s2 += red[1][k][cl];
    }
    const double mean = s1 / P;
    double var = s2 / P - mean * mean;
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const float sc = gamma[c] * rstd;
    save_mean[c] = (float)mean;
    save_rstd[c] = rstd;
    scale[c] = sc;
    shift[c] = fmaf(-(float)mean, sc, beta[c]);
    if (running_mean != nullptr) {
      const double unb = P > 1 ? var * ((double)P / (double)(P - 1)) : var;
      // the statistics are those of the bias-free conv output; the reference's BN sees conv + bias
      const float mean_b = (float)mean + (conv_bias != nullptr ? conv_bias[c] : 0.f);
      running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean_b;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
    }
  }
}

int tbn_launch_bn_finalize(const float* partial, int nparts, int P, int C, const float* gamma, const float* beta,
                           const float* conv_bias, float* running_mean, float* running_var, float momentum, float eps,
                           float* save_mean, float* save_rstd, float* scale, float* shift, hipStream_t st) {
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(C, 8)), dim3(256), 0, st, partial, nparts, P, C, gamma, beta,
                     conv_bias, running_mean, running_var, momentum, eps, save_mean, save_rstd, scale, shift);
  TBN_CHECK_LAUNCH("bn_finalize");
  return TBN_OK;
}

// ---------------------------------------------------------------- apply: z = relu(y*scale+shift) -> concat slices
struct Seg3 {
  Seg s[3];
  int n;
};
struct CSeg3 {
  CSeg s[3];
  int n;
};

__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ y, int P, int C,
                                                       const float* __restrict__ scale,
                                                       const float* __restrict__ shift, Seg3 segs) {
  const int G = C >> 2;
  const size_t total = (size_t)P * G;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int p = (int)(i / G), c = (int)(i - (size_t)p * G) * 4;
    const float4 v = *reinterpret_cast<const float4*>(y + (size_t)p * C + c);
    const float4 sc = *reinterpret_cast<const float4*>(scale + c);
    const float4 sh = *reinterpret_cast<const float4*>(shift + c);
    float4 z;
    z.x = fmaxf(fmaf(v.x, sc.x, sh.x), 0.f);
    z.y = fmaxf(fmaf(v.y, sc.y, sh.y), 0.f);
    z.z = fmaxf(fmaf(v.z, sc.z, sh.z), 0.f);
    z.w = fmaxf(fmaf(v.w, sc.w, sh.w), 0.f);
    int sg = 0;
    if (segs.n > 1 && c >= segs.s[1].col_begin) sg = 1;
    if (segs.n > 2 && c >= segs.s[2].col_begin) sg = 2;
    *reinterpret_cast<float4*>(segs.s[sg].ptr + (size_t)p * segs.s[sg].ld + (c - segs.s[sg].col_begin)) = z;
  }
}

static inline int ew_grid(size_t items) {
  size_t g = (items + 255) / 256;
  return (int)(g > 4096 ? 4096 : (g < 1 ? 1 : g));
}

int tbn_launch_bn_apply(const float* y, int P, int C, const float* scale, const float* shift, const Seg* segs,
                        int nseg, hipStream_t st) {
  TBN_REQUIRE(C % 4 == 0 && nseg >= 1 && nseg <= 3, "bn_apply: bad C/nseg");
  Seg3 s3;
  s3.n = nseg;
  for (int i = 0; i < nseg; ++i) {
    s3.s[i] = segs[i];
    TBN_REQUIRE(segs[i].ld % 4 == 0 && segs[i].col_begin % 4 == 0, "bn_apply: segment pitch/offset must be x4");
  }
  hipLaunchKernelGGL(bn_apply_kernel, dim3(ew_grid((size_t)P * C / 4)), dim3(256), 0, st, y, P, C, scale, shift, s3);
  TBN_CHECK_LAUNCH("bn_apply");
  return TBN_OK;
}

// ---------------------------------------------------------------- eval fold: running stats -> scale/shift
// scale = gamma / sqrt(var + eps);  shift = beta + (conv_bias - mean) * scale   (the conv epilogue
// then computes relu(acc * scale + shift) on the bias-free accumulator)
__global__ void bn_fold_kernel(const float* gamma, const float* beta, const float* mean, const float* var,
                               const float* bias, float eps, float* scale, float* shift, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) {
    const float sc = gamma[c] / sqrtf(var[c] + eps);
    scale[c] = sc;
    shift[c] = fmaf((bias != nullptr ? bias[c] : 0.f) - mean[c], sc, beta[c]);
  }
}

int tbn_launch_bn_fold(const float* gamma, const float* beta, const float* mean, const float* var, const float* bias,
                       float eps, float* scale, float* shift, int C, hipStream_t st) {
  hipLaunchKernelGGL(bn_fold_kernel, dim3(cdiv(C, 256)), dim3(256), 0, st, gamma, beta, mean, var, bias, eps, scale,
                     shift, C);
  TBN_CHECK_LAUNCH("bn_fold");
  return TBN_OK;
}

// ---------------------------------------------------------------- backward
// g = dz * [y*scale+shift > 0];  xhat = (y-mean)*rstd;  S1 = sum g, S2 = sum g*xhat
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(CSeg3 dz, const float* __restrict__ y, int P, int C,
                                                            int pch, const float* __restrict__ scale,
                                                            const float* __restrict__ shift,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ rstd,
                                                            float* __restrict__ partial) {
  __shared__ float red[2 * 2048];
  const int G = C >> 2, RP = 256 / G;
  const int tid = threadIdx.x, cg = tid % G, rs = tid / G, c = cg * 4;
  const int p0 = blockIdx.x * pch, p1 = min(P, p0 + pch);
  float4 s1 = make_float4(0, 0, 0, 0), s2 = s1;
  if (rs < RP) {
    int sg = 0;
    if (dz.n > 1 && c >= dz.s[1].col_begin) sg = 1;
    if (dz.n > 2 && c >= dz.s[2].col_begin) sg = 2;
    const float* dzp = dz.s[sg].ptr + (c - dz.s[sg].col_begin);
    const int dld = dz.s[sg].ld;
    const float4 sc = *reinterpret_cast<const float4*>(scale + c);
    const float4 sh = *reinterpret_cast<const float4*>(shift + c);
    const float4 mu = *reinterpret_cast<const float4*>(mean + c);
    const float4 rs4 = *reinterpret_cast<const float4*>(rstd + c);
    for (int p = p0 + rs; p < p1; p += RP) {
      const float4 v = *reinterpret_cast<const float4*>(y + (size_t)p * C + c);
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
    float a = 0.f, b = 0.f;
    for (int r = 0; r < RP; ++r) {
      a += red[r * C + cc];
      b += red[2048 + r * C + cc];
    }
    partial[((size_t)blockIdx.x * 2 + 0) * C + cc] = a;
    partial[((size_t)blockIdx.x * 2 + 1) * C + cc] = b;
  }
}

int tbn_bn_bwd_parts(int P, int C) { return tbn_bn_stats_parts(P, C); }

static int fill_cseg3(CSeg3* o, const CSeg* s, int n) {
  o->n = n;
  for (int i = 0; i < n; ++i) {
    o->s[i] = s[i];
    if (s[i].ld % 4 != 0 || s[i].col_begin % 4 != 0) return -1;
  }
  return 0;
}

int tbn_launch_bn_bwd_reduce(const CSeg* dz, int nseg, const float* y, int P, int C, const float* scale,
                             const float* shift, const float* mean, const float* rstd, float* partial,
                             hipStream_t st) {
  TBN_REQUIRE(C % 4 == 0 && C <= 1024 && nseg >= 1 && nseg <= 3, "bn_bwd_reduce: bad C/nseg");
  CSeg3 s3;
  TBN_REQUIRE(fill_cseg3(&s3, dz, nseg) == 0, "bn_bwd_reduce: segment pitch/offset must be x4");
  const int rp = 256 / (C / 4), pch = pick_chunk(P, rp), parts = cdiv(P, pch);
  hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(parts), dim3(256), 0, st, s3, y, P, C, pch, scale, shift, mean, rstd,
                     partial);
  TBN_CHECK_LAUNCH("bn_bwd_reduce");
  return TBN_OK;
}

// coef[0][c]=a, coef[1][c]=b, coef[2][c]=cst with dy = a*g + b*y + cst
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* __restrict__ partial, int nparts, int P,
                                                              int C, const float* __restrict__ scale,
                                                              const float* __restrict__ mean,
                                                              const float* __restrict__ rstd, float* coef,
                                                              float* dgamma, float* dbeta, float* dbias) {
  __shared__ double red[2][32][8];
  const int tid = threadIdx.x, cl = tid & 7, slot = tid >> 3;
  const int c = blockIdx.x * 8 + cl;
  double a = 0.0, b = 0.0;
  if (c < C)
    for (int i = slot; i < nparts; i += 32) {
      a += (double)partial[((size_t)i * 2 + 0) * C + c];
      b += (double)partial[((size_t)i * 2 + 1) * C + c];
    }
  red[0][slot][cl] = a;
  red[1][slot][cl] = b;
  __syncthreads();
  if (slot == 0 && c < C) {
    double s1 = 0.0, s2 = 0.0;
    for (int k = 0; k < 32; ++k) {
      s1 += red[0][k][cl];
      s2 += red[1][k][cl];
    }
    const double sc = scale[c], rs = rstd[c], mu = mean[c];
    const double bb = -sc * rs * (s2 / P);
    coef[c] = (float)sc;
    coef[C + c] = (float)bb;
    coef[2 * C + c] = (float)(-sc * (s1 / P) - bb * mu);
    if (dgamma) dgamma[c] = (float)s2;
    if (dbeta) dbeta[c] = (float)s1;
    // a per-channel constant added before a batch-stat BN has exactly zero gradient
    if (dbias) dbias[c] = 0.f;
  }
}

int tbn_launch_bn_bwd_finalize(const float* partial, int nparts, int P, int C, const float* scale, const float* mean,
                               const float* rstd, float* coef, float* dgamma, float* dbeta, float* dbias,
                               hipStream_t st) {
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(cdiv(C, 8)), dim3(256), 0, st, partial, nparts, P, C, scale, mean,
                     rstd, coef, dgamma, dbeta, dbias);
  TBN_CHECK_LAUNCH("bn_bwd_finalize");
  return TBN_OK;
}

// y and dy may alias (the engine converts y to dy in place): no __restrict__ on them
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(CSeg3 dz, const float* y, int P, int C,
                                                           const float* __restrict__ scale,
                                                           const float* __restrict__ shift,
                                                           const float* __restrict__ coef, float* dy) {
  const int G = C >> 2;
  const size_t total = (size_t)P * G;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int p = (int)(i / G), c = (int)(i - (size_t)p * G) * 4;
    int sg = 0;
    if (dz.n > 1 && c >= dz.s[1].col_begin) sg = 1;
    if (dz.n > 2 && c >= dz.s[2].col_begin) sg = 2;
    const float4 d = *reinterpret_cast<const float4*>(dz.s[sg].ptr + (size_t)p * dz.s[sg].ld + (c - dz.s[sg].col_begin));
    const float4 v = *reinterpret_cast<const float4*>(y + (size_t)p * C + c);
    const float4 sc = *reinterpret_cast<const float4*>(scale + c);
    const float4 sh = *reinterpret_cast<const float4*>(shift + c);
    const float4 ca = *reinterpret_cast<const float4*>(coef + c);
    const float4 cb = *reinterpret_cast<const float4*>(coef + C + c);
    const float4 cc = *reinterpret_cast<const float4*>(coef + 2 * C + c);
    float4 o;
    o.x = fmaf(ca.x, fmaf(v.x, sc.x, sh.x) > 0.f ? d.x : 0.f, fmaf(cb.x, v.x, cc.x));
    o.y = fmaf(ca.y, fmaf(v.y, sc.y, sh.y) > 0.f ? d.y : 0.f, fmaf(cb.y, v.y, cc.y));
    o.z = fmaf(ca.z, fmaf(v.z, sc.z, sh.z) > 0.f ? d.z : 0.f, fmaf(cb.z, v.z, cc.z));
    o.w = fmaf(ca.w, fmaf(v.w, sc.w, sh.w) > 0.f ? d.w : 0.f, fmaf(cb.w, v.w, cc.w));
    *reinterpret_cast<float4*>(dy + (size_t)p * C + c) = o;
  }
}

int tbn_launch_bn_bwd_apply(const CSeg* dz, int nseg, const float* y, int P, int C, const float* scale,
                            const float* shift, const float* coef, float* dy, hipStream_t st) {
  CSeg3 s3;
  TBN_REQUIRE(C % 4 == 0 && nseg >= 1 && nseg <= 3 && fill_cseg3(&s3, dz, nseg) == 0, "bn_bwd_apply: bad segments");
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(ew_grid((size_t)P * C / 4)), dim3(256), 0, st, s3, y, P, C, scale,
                     shift, coef, dy);
  TBN_CHECK_LAUNCH("bn_bwd_apply");
  return TBN_OK;
}
