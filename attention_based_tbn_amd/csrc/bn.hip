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
#include "tbn_bn_dev.h"
#include "tbn_pool_dev.h"

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
  TBN_KLAUNCH(bn_stats_kernel, dim3(parts), dim3(256), 0, st, y, ld, P, C, pch, partial);
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
  // 16 channels per workgroup, fixed summation order (tbn_bn_dev.h)
  __shared__ double red[64 * 32];
  const int tid = threadIdx.x;
  const int c = blockIdx.x * TBN_FIN_CH + tid;
  double s1, s2;
  tbn_sum_partials<TBN_FIN_CH>(partial, C, nparts, blockIdx.x * TBN_FIN_CH, C, red, &s1, &s2);
  if (tid < TBN_FIN_CH && c < C) {
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
  TBN_KLAUNCH(bn_finalize_kernel, dim3(cdiv(C, TBN_FIN_CH)), dim3(256), 0, st, partial, nparts, P, C, gamma, beta,
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
                                                       const float* __restrict__ shift, Seg3 segs, FastDiv divg) {
  const int G = C >> 2;
  const uint32_t total = (uint32_t)P * G;
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
    const int p = (int)fdiv(i, divg), c = (int)(i - (uint32_t)p * (uint32_t)G) * 4;
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
  TBN_REQUIRE((size_t)P * (C / 4) < (1ull << 31), "bn_apply: too many elements per call");
  TBN_KLAUNCH(bn_apply_kernel, dim3(ew_grid((size_t)P * C / 4)), dim3(256), 0, st, y, P, C, scale, shift, s3,
                     make_fastdiv((uint32_t)(C / 4)));
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
  TBN_KLAUNCH(bn_fold_kernel, dim3(cdiv(C, 256)), dim3(256), 0, st, gamma, beta, mean, var, bias, eps, scale,
                     shift, C);
  TBN_CHECK_LAUNCH("bn_fold");
  return TBN_OK;
}

// ---------------------------------------------------------------- apply + 3x3 max pool in one pass
// For a conv whose BN-ReLU output is consumed ONLY by a max pool (the stem: conv1 -> pool1, conv2_3x3 -> pool2):
// z = relu(y*scale+shift) is formed in registers and pooled at once; z itself is never written (the backward
// recomputes the ReLU mask from y anyway).  Same comparison order as maxpool_fwd_kernel on a stored z (first
// maximum in scan order, NaN propagates), so values and arg-max are identical to the two-pass form.
__global__ __launch_bounds__(256) void bn_apply_maxpool_kernel(const float* __restrict__ y, int N, int H, int W, int C,
                                                               const float* __restrict__ scale,
                                                               const float* __restrict__ shift,
                                                               float* __restrict__ out, int out_ld,
                                                               unsigned char* __restrict__ argmax, int OH, int OW,
                                                               int stride, int pad, PixDecode dec) {
  const int G = C >> 2;
  const uint32_t total = (uint32_t)N * OH * OW * G;
  const unsigned nb = gridDim.x, bid = blockIdx.x;   // XCD-aware order: a window row is fetched into one L2
  const unsigned q8 = nb >> 3, r8 = nb & 7, xcd = bid & 7, idx = bid >> 3;
  const uint32_t b0 = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
  for (uint32_t i = b0 * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
    int g, ox, oy, n;
    uint32_t opix;
    pix_decode(dec, i, g, ox, oy, n, opix);
    const int y0 = oy * stride - pad, x0 = ox * stride - pad;
    const float4 sc = *reinterpret_cast<const float4*>(scale + g * 4);
    const float4 sh = *reinterpret_cast<const float4*>(shift + g * 4);
    float4 best = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    int bx = 0, by = 0, bz = 0, bw = 0;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int s2 = 0; s2 < 3; ++s2) {
        const int iy = y0 + r, ix = x0 + s2;
        if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) {
          const float4 v = *reinterpret_cast<const float4*>(y + ((size_t)(n * H + iy) * W + ix) * C + g * 4);
          float4 z;
          z.x = fmaxf(fmaf(v.x, sc.x, sh.x), 0.f);
          z.y = fmaxf(fmaf(v.y, sc.y, sh.y), 0.f);
          z.z = fmaxf(fmaf(v.z, sc.z, sh.z), 0.f);
          z.w = fmaxf(fmaf(v.w, sc.w, sh.w), 0.f);
          const int k = r * 3 + s2;
          if (z.x > best.x || z.x != z.x) { best.x = z.x; bx = k; }
          if (z.y > best.y || z.y != z.y) { best.y = z.y; by = k; }
          if (z.z > best.z || z.z != z.z) { best.z = z.z; bz = k; }
          if (z.w > best.w || z.w != z.w) { best.w = z.w; bw = k; }
        }
      }
    *reinterpret_cast<float4*>(out + (size_t)opix * out_ld + g * 4) = best;
    *reinterpret_cast<uint32_t*>(argmax + (size_t)opix * C + g * 4) =
        (uint32_t)bx | ((uint32_t)by << 8) | ((uint32_t)bz << 16) | ((uint32_t)bw << 24);
  }
}

int tbn_launch_bn_apply_maxpool(const float* y, int N, int H, int W, int C, const float* scale, const float* shift,
                                float* out, int out_ld, unsigned char* argmax, int OH, int OW, int stride, int pad,
                                hipStream_t st) {
  TBN_REQUIRE(C % 4 == 0 && out_ld % 4 == 0 && argmax != nullptr, "bn_apply_maxpool: bad C / pitch / argmax");
  size_t g = ((size_t)N * OH * OW * (C / 4) + 255) / 256;
  if (g > 8192) g = 8192;
  TBN_REQUIRE((size_t)N * H * W * (C / 4) < (1ull << 31), "bn_apply_maxpool: too many elements per call");
  TBN_KLAUNCH(bn_apply_maxpool_kernel, dim3((unsigned)(g < 1 ? 1 : g)), dim3(256), 0, st, y, N, H, W, C, scale,
                     shift, out, out_ld, argmax, OH, OW, stride, pad, make_pixdecode(C / 4, OW, OH));
  TBN_CHECK_LAUNCH("bn_apply_maxpool");
  return TBN_OK;
}

// Gradient of a max pool's INPUT formed on the fly from the pooled gradient and the stored arg-max (the gather of
// maxpool_bwd_kernel): used by the BN-backward kernels of a conv fused with its pool, so dz is never materialised.
struct PoolGrad {
  const float* dout;            // gradient of the pooled tensor (N, OH, OW, C) with pitch dout_ld
  const unsigned char* argmax;  // (N, OH, OW, C) uint8 window index
  int dout_ld, H, W, OH, OW, C, stride, pad;
  FastDiv div_w, div_h;
};
__device__ __forceinline__ float4 pooled_grad(const PoolGrad& q, int p, int c) {
  const uint32_t row = fdiv((uint32_t)p, q.div_w);        // n*H + iy
  const int ix = p - (int)row * q.W;
  const uint32_t n = fdiv(row, q.div_h);
  const int iy = (int)row - (int)n * q.H;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (q.stride == 2 && q.pad == 0) {
    // 3x3 / stride 2 / pad 0 (every fused pool): an input pixel lies in window o = i >> 1 (tap i & 1) and, for even
    // i, also in window o - 1 (tap 2).  All four candidates are evaluated branch-free with their loads issued
    // together (the looped form below is latency-bound: 1.3 TB/s on the 308 MB stem tensors); invalid candidates
    // read a clamped, in-range address and are masked.  Same summation order as the loops: (hi,hi) (hi,lo) (lo,hi) (lo,lo).
    const int oyA = iy >> 1, ryA = iy & 1, oxA = ix >> 1, rxA = ix & 1;
    const bool vyA = oyA < q.OH, vyB = (ryA == 0) && (oyA >= 1);
    const bool vxA = oxA < q.OW, vxB = (rxA == 0) && (oxA >= 1);
    const int oy[2] = {vyA ? oyA : 0, vyB ? oyA - 1 : 0}, ry[2] = {ryA, 2};
    const int ox[2] = {vxA ? oxA : 0, vxB ? oxA - 1 : 0}, rx[2] = {rxA, 2};
    const bool vy[2] = {vyA, vyB}, vx[2] = {vxA, vxB};
    uint32_t am[4];
    float4 d[4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const size_t opix = (size_t)((int)n * q.OH + oy[a]) * q.OW + ox[b];
        am[a * 2 + b] = *reinterpret_cast<const uint32_t*>(q.argmax + opix * q.C + c);
        d[a * 2 + b] = *reinterpret_cast<const float4*>(q.dout + opix * q.dout_ld + c);
      }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const uint32_t k = (uint32_t)(ry[a] * 3 + rx[b]);
        const uint32_t m = am[a * 2 + b];
        const bool v = vy[a] && vx[b];
        if (v && (m & 0xff) == k) acc.x += d[a * 2 + b].x;
        if (v && ((m >> 8) & 0xff) == k) acc.y += d[a * 2 + b].y;
        if (v && ((m >> 16) & 0xff) == k) acc.z += d[a * 2 + b].z;
        if (v && (m >> 24) == k) acc.w += d[a * 2 + b].w;
      }
    return acc;
  }
  const int oy_hi = min(q.OH - 1, (iy + q.pad) / q.stride);
  const int ox_hi = min(q.OW - 1, (ix + q.pad) / q.stride);
  for (int oy = oy_hi; oy >= 0; --oy) {
    const int r = iy - (oy * q.stride - q.pad);
    if (r > 2) break;
    for (int ox = ox_hi; ox >= 0; --ox) {
      const int s = ix - (ox * q.stride - q.pad);
      if (s > 2) break;
      const uint32_t k = (uint32_t)(r * 3 + s);
      const size_t opix = (size_t)((int)n * q.OH + oy) * q.OW + ox;
      const uint32_t am = *reinterpret_cast<const uint32_t*>(q.argmax + opix * q.C + c);
      const float4 d = *reinterpret_cast<const float4*>(q.dout + opix * q.dout_ld + c);
      if ((am & 0xff) == k) acc.x += d.x;
      if (((am >> 8) & 0xff) == k) acc.y += d.y;
      if (((am >> 16) & 0xff) == k) acc.z += d.z;
      if ((am >> 24) == k) acc.w += d.w;
    }
  }
  return acc;
}

// ---------------------------------------------------------------- backward
// g = dz * [y*scale+shift > 0];  xhat = (y-mean)*rstd;  S1 = sum g, S2 = sum g*xhat
template <bool POOLED>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(CSeg3 dz, PoolGrad pg, const float* __restrict__ y, int P,
                                                            int C, int pch, const float* __restrict__ scale,
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
    const float* dzp = POOLED ? nullptr : dz.s[sg].ptr + (c - dz.s[sg].col_begin);
    const int dld = dz.s[sg].ld;
    const float4 sc = *reinterpret_cast<const float4*>(scale + c);
    const float4 sh = *reinterpret_cast<const float4*>(shift + c);
    const float4 mu = *reinterpret_cast<const float4*>(mean + c);
    const float4 rs4 = *reinterpret_cast<const float4*>(rstd + c);
    for (int p = p0 + rs; p < p1; p += RP) {
      const float4 v = *reinterpret_cast<const float4*>(y + (size_t)p * C + c);
      const float4 d = POOLED ? pooled_grad(pg, p, c) : *reinterpret_cast<const float4*>(dzp + (size_t)p * dld);
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
  PoolGrad none = {};
  TBN_KLAUNCH(bn_bwd_reduce_kernel<false>, dim3(parts), dim3(256), 0, st, s3, none, y, P, C, pch, scale, shift,
                     mean, rstd, partial);
  TBN_CHECK_LAUNCH("bn_bwd_reduce");
  return TBN_OK;
}

static PoolGrad make_poolgrad(const float* dout, int dout_ld, const unsigned char* argmax, int H, int W, int OH, int OW,
                              int C, int stride, int pad) {
  PoolGrad q;
  q.dout = dout; q.argmax = argmax; q.dout_ld = dout_ld;
  q.H = H; q.W = W; q.OH = OH; q.OW = OW; q.C = C; q.stride = stride; q.pad = pad;
  q.div_w = make_fastdiv((uint32_t)W);
  q.div_h = make_fastdiv((uint32_t)H);
  return q;
}

// ---------------------------------------------------------------- fused-pool backward on 2x2 input blocks
// 3x3 / stride 2 / pad 0 (the stem pools: the only fused ones).  pooled_grad() above evaluates four candidate windows
// for every input pixel; taken as a 2x2 block (even-even, even-odd, odd-even, odd-odd pixel) the four pixels share the
// SAME four windows -- (by, bx), (by, bx-1), (by-1, bx), (by-1, bx-1) -- and only 9 (pixel, window) pairs exist:
// one arg-max word + one gradient quad per window are loaded once per block instead of four per pixel (4x fewer
// loads, ~2x fewer VALU instructions on the two largest tensors of the network).  Same per-pixel summation order.
__global__ __launch_bounds__(256) void bn_bwd_reduce_pooled2x2_kernel(PoolBlk pb, const float* __restrict__ y, int NQ,
                                                                       int C, int qch, const float* __restrict__ scale,
                                                                       const float* __restrict__ shift,
                                                                       const float* __restrict__ mean,
                                                                       const float* __restrict__ rstd,
                                                                       float* __restrict__ partial) {
  __shared__ float red[2 * 2048];
  const int G = C >> 2, RP = 256 / G;
  const int tid = threadIdx.x, cg = tid % G, rs = tid / G, c = cg * 4;
  const int q0 = blockIdx.x * qch, q1 = min(NQ, q0 + qch);
  float4 s1 = make_float4(0, 0, 0, 0), s2 = s1;
  if (rs < RP) {
    const float4 sc = *reinterpret_cast<const float4*>(scale + c);
    const float4 sh = *reinterpret_cast<const float4*>(shift + c);
    const float4 mu = *reinterpret_cast<const float4*>(mean + c);
    const float4 rs4 = *reinterpret_cast<const float4*>(rstd + c);
    for (int q = q0 + rs; q < q1; q += RP) {
      const uint32_t row = fdiv((uint32_t)q, pb.div_bw);       // n*BH + by
      const int bx = q - (int)row * pb.BW;
      const uint32_t n = fdiv(row, pb.div_bh);
      const int by = (int)row - (int)n * pb.BH;
      float4 g[4];
      pooled_grad_2x2(pb, (int)n, by, bx, c, g);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int iy = 2 * by + (i >> 1), ix = 2 * bx + (i & 1);
        if (iy < pb.H && ix < pb.W) {
          const float4 v = *reinterpret_cast<const float4*>(y + ((size_t)((int)n * pb.H + iy) * pb.W + ix) * C + c);
          const float gx = fmaf(v.x, sc.x, sh.x) > 0.f ? g[i].x : 0.f;
          const float gy = fmaf(v.y, sc.y, sh.y) > 0.f ? g[i].y : 0.f;
          const float gz = fmaf(v.z, sc.z, sh.z) > 0.f ? g[i].z : 0.f;
          const float gw = fmaf(v.w, sc.w, sh.w) > 0.f ? g[i].w : 0.f;
          s1.x += gx; s1.y += gy; s1.z += gz; s1.w += gw;
          s2.x = fmaf(gx, (v.x - mu.x) * rs4.x, s2.x);
          s2.y = fmaf(gy, (v.y - mu.y) * rs4.y, s2.y);
          s2.z = fmaf(gz, (v.z - mu.z) * rs4.z, s2.z);
          s2.w = fmaf(gw, (v.w - mu.w) * rs4.w, s2.w);
        }
      }
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

__global__ __launch_bounds__(256) void bn_bwd_apply_pooled2x2_kernel(PoolBlk pb, const float* y, int NQ, int C,
                                                                      const float* __restrict__ scale,
                                                                      const float* __restrict__ shift,
                                                                      const float* __restrict__ coef, float* dy,
                                                                      FastDiv divg) {
  const int G = C >> 2;
  const uint32_t total = (uint32_t)NQ * G;
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
    const uint32_t q = fdiv(i, divg);
    const int c = (int)(i - q * (uint32_t)G) * 4;
    const uint32_t row = fdiv(q, pb.div_bw);
    const int bx = (int)q - (int)row * pb.BW;
    const uint32_t n = fdiv(row, pb.div_bh);
    const int by = (int)row - (int)n * pb.BH;
    float4 g[4];
    pooled_grad_2x2(pb, (int)n, by, bx, c, g);
    const float4 sc = *reinterpret_cast<const float4*>(scale + c);
    const float4 sh = *reinterpret_cast<const float4*>(shift + c);
    const float4 ca = *reinterpret_cast<const float4*>(coef + c);
    const float4 cb = *reinterpret_cast<const float4*>(coef + C + c);
    const float4 cc = *reinterpret_cast<const float4*>(coef + 2 * C + c);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int iy = 2 * by + (k >> 1), ix = 2 * bx + (k & 1);
      if (iy < pb.H && ix < pb.W) {
        const size_t off = ((size_t)((int)n * pb.H + iy) * pb.W + ix) * C + c;
        const float4 v = tbn_ld4<(TBN_BN_NT & 8) != 0>(y + off);
        float4 o;
        o.x = fmaf(ca.x, fmaf(v.x, sc.x, sh.x) > 0.f ? g[k].x : 0.f, fmaf(cb.x, v.x, cc.x));
        o.y = fmaf(ca.y, fmaf(v.y, sc.y, sh.y) > 0.f ? g[k].y : 0.f, fmaf(cb.y, v.y, cc.y));
        o.z = fmaf(ca.z, fmaf(v.z, sc.z, sh.z) > 0.f ? g[k].z : 0.f, fmaf(cb.z, v.z, cc.z));
        o.w = fmaf(ca.w, fmaf(v.w, sc.w, sh.w) > 0.f ? g[k].w : 0.f, fmaf(cb.w, v.w, cc.w));
        *reinterpret_cast<float4*>(dy + off) = o;
      }
    }
  }
}

// number of partial rows the pooled reduce writes (the engine sizes the finalize by it)
int tbn_bn_bwd_pooled_parts(int N, int H, int W, int C, int stride, int pad) {
  if (stride == 2 && pad == 0) {
    const int NQ = N * ((H + 1) / 2) * ((W + 1) / 2);
    const int rp = 256 / (C / 4);
    return cdiv(NQ, pick_chunk(NQ, rp));
  }
  return tbn_bn_bwd_parts(N * H * W, C);
}

// BN-backward reduce / apply of a conv fused with its max pool: dz = gather(dpooled, argmax) on the fly
int tbn_launch_bn_bwd_reduce_pooled(const float* dpooled, int dpooled_ld, const unsigned char* argmax, int N, int H,
                                    int W, int OH, int OW, int stride, int pad, const float* y, int C,
                                    const float* scale, const float* shift, const float* mean, const float* rstd,
                                    float* partial, hipStream_t st) {
  TBN_REQUIRE(C % 4 == 0 && C <= 1024 && dpooled_ld % 4 == 0, "bn_bwd_reduce_pooled: bad C / pitch");
  TBN_REQUIRE((long)N * H * W < (1l << 31), "bn_bwd_reduce_pooled: too many pixels");
  if (stride == 2 && pad == 0) {
    const int NQ = N * ((H + 1) / 2) * ((W + 1) / 2);
    const int rp = 256 / (C / 4), qch = pick_chunk(NQ, rp), parts = cdiv(NQ, qch);
    TBN_KLAUNCH(bn_bwd_reduce_pooled2x2_kernel, dim3(parts), dim3(256), 0, st,
                       make_poolblk(dpooled, dpooled_ld, argmax, H, W, OH, OW, C), y, NQ, C, qch, scale, shift, mean, rstd,
                       partial);
    TBN_CHECK_LAUNCH("bn_bwd_reduce_pooled2x2");
    return TBN_OK;
  }
  const int P = N * H * W;
  CSeg3 s3 = {};
  s3.n = 1;
  const int rp = 256 / (C / 4), pch = pick_chunk(P, rp), parts = cdiv(P, pch);
  TBN_KLAUNCH(bn_bwd_reduce_kernel<true>, dim3(parts), dim3(256), 0, st, s3,
                     make_poolgrad(dpooled, dpooled_ld, argmax, H, W, OH, OW, C, stride, pad), y, P, C, pch, scale,
                     shift, mean, rstd, partial);
  TBN_CHECK_LAUNCH("bn_bwd_reduce_pooled");
  return TBN_OK;
}

// coef[0][c]=a, coef[1][c]=b, coef[2][c]=cst with dy = a*g + b*y + cst
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* __restrict__ partial, int nparts, int P,
                                                              int C, const float* __restrict__ scale,
                                                              const float* __restrict__ mean,
                                                              const float* __restrict__ rstd, float* coef,
                                                              float* dgamma, float* dbeta, float* dbias) {
  __shared__ double red[64 * 32];
  const int tid = threadIdx.x;
  const int c = blockIdx.x * TBN_FIN_CH + tid;
  double s1, s2;
  tbn_sum_partials<TBN_FIN_CH>(partial, C, nparts, blockIdx.x * TBN_FIN_CH, C, red, &s1, &s2);
  if (tid < TBN_FIN_CH && c < C) {
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
  TBN_KLAUNCH(bn_bwd_finalize_kernel, dim3(cdiv(C, TBN_FIN_CH)), dim3(256), 0, st, partial, nparts, P, C, scale, mean,
                     rstd, coef, dgamma, dbeta, dbias);
  TBN_CHECK_LAUNCH("bn_bwd_finalize");
  return TBN_OK;
}

// y and dy may alias (the engine converts y to dy in place): no __restrict__ on them
template <bool POOLED>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(CSeg3 dz, PoolGrad pg, const float* y, int P, int C,
                                                           const float* __restrict__ scale,
                                                           const float* __restrict__ shift,
                                                           const float* __restrict__ coef, float* dy, FastDiv divg) {
  const int G = C >> 2;
  const uint32_t total = (uint32_t)P * G;
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
    const int p = (int)fdiv(i, divg), c = (int)(i - (uint32_t)p * (uint32_t)G) * 4;
    int sg = 0;
    if (dz.n > 1 && c >= dz.s[1].col_begin) sg = 1;
    if (dz.n > 2 && c >= dz.s[2].col_begin) sg = 2;
    const float4 d = POOLED ? pooled_grad(pg, p, c)
                            : *reinterpret_cast<const float4*>(dz.s[sg].ptr + (size_t)p * dz.s[sg].ld +
                                                               (c - dz.s[sg].col_begin));
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
  PoolGrad none = {};
  TBN_REQUIRE((size_t)P * (C / 4) < (1ull << 31), "bn_bwd_apply: too many elements per call");
  TBN_KLAUNCH(bn_bwd_apply_kernel<false>, dim3(ew_grid((size_t)P * C / 4)), dim3(256), 0, st, s3, none, y, P, C,
                     scale, shift, coef, dy, make_fastdiv((uint32_t)(C / 4)));
  TBN_CHECK_LAUNCH("bn_bwd_apply");
  return TBN_OK;
}

int tbn_launch_bn_bwd_apply_pooled(const float* dpooled, int dpooled_ld, const unsigned char* argmax, int N, int H, int W,
                                   int OH, int OW, int stride, int pad, const float* y, int C, const float* scale,
                                   const float* shift, const float* coef, float* dy, hipStream_t st) {
  TBN_REQUIRE(C % 4 == 0 && dpooled_ld % 4 == 0, "bn_bwd_apply_pooled: bad C / pitch");
  if (stride == 2 && pad == 0) {
    const int NQ = N * ((H + 1) / 2) * ((W + 1) / 2);
    TBN_REQUIRE((size_t)NQ * (C / 4) < (1ull << 31), "bn_bwd_apply_pooled: too many elements per call");
    TBN_KLAUNCH(bn_bwd_apply_pooled2x2_kernel, dim3(ew_grid((size_t)NQ * C / 4)), dim3(256), 0, st,
                       make_poolblk(dpooled, dpooled_ld, argmax, H, W, OH, OW, C), y, NQ, C, scale, shift, coef, dy,
                       make_fastdiv((uint32_t)(C / 4)));
    TBN_CHECK_LAUNCH("bn_bwd_apply_pooled2x2");
    return TBN_OK;
  }
  const int P = N * H * W;
  CSeg3 s3 = {};
  s3.n = 1;
  TBN_KLAUNCH(bn_bwd_apply_kernel<true>, dim3(ew_grid((size_t)P * C / 4)), dim3(256), 0, st, s3,
                     make_poolgrad(dpooled, dpooled_ld, argmax, H, W, OH, OW, C, stride, pad), y, P, C, scale, shift,
                     coef, dy, make_fastdiv((uint32_t)(C / 4)));
  TBN_CHECK_LAUNCH("bn_bwd_apply_pooled");
  return TBN_OK;
}
