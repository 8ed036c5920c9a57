// Mid-level fusion heads for gfx950: positional-encoding concat, GroupNorm, the L_q = 1
// multi-head attention core (one 64-lane wave per (sample, head): dot products and the softmax
// are wavefront reductions), fixed-attention weighted sum, temporal-consensus mean.
//
// Reference: core/models/attention.py:8-57, core/models/model.py:62-67,178-203,224-237.
// All tensors are row-major with channels fastest: audio sequence (r, t, c), features (r, c).
#include "tbn_common.h"
#include "../../include/tbn_hip.h"

static inline int ew_grid(size_t items) {
  size_t g = (items + 255) / 256;
  return (int)(g > 4096 ? 4096 : (g < 1 ? 1 : g));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// ---------------------------------------------------------------- PE concat
__global__ __launch_bounds__(256) void pe_concat_kernel(const float* __restrict__ feat, int feat_ld,
                                                        const float* __restrict__ pe, float* __restrict__ out,
                                                        int out_ld, int R, int T, int C, int PD) {
  const size_t total = (size_t)R * T * out_ld;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int c = (int)(i % out_ld);
    const size_t row = i / out_ld;
    const int t = (int)(row % T);
    float v = 0.f;
    if (c < C)
      v = feat[row * feat_ld + c];
    else if (c < C + PD)
      v = pe[(c - C) * T + t];
    out[i] = v;
  }
}

// ---------------------------------------------------------------- GroupNorm over (t, c/groups) per sample
// one workgroup per sample; thread i owns channels 4i..4i+3 for all t; a group (16 ch) = 4 lanes
__global__ __launch_bounds__(256) void groupnorm_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float* save_mean,
                                                            float* save_rstd, int T, int C, int groups, float eps) {
  const int r = blockIdx.x, tid = threadIdx.x;
  const int cpg = C / groups, lanes = cpg / 4;  // lanes per group (power of two, <= 64)
  for (int c = tid * 4; c < C; c += 1024) {
    float s1 = 0.f, s2 = 0.f;
    for (int t = 0; t < T; ++t) {
      const float4 v = *reinterpret_cast<const float4*>(x + ((size_t)r * T + t) * C + c);
      s1 += (v.x + v.y) + (v.z + v.w);
      s2 += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    }
    for (int o = 1; o < lanes; o <<= 1) {
      s1 += __shfl_xor(s1, o);
      s2 += __shfl_xor(s2, o);
    }
    const float n = (float)(cpg * T);
    const float mean = s1 / n;
    const float var = fmaxf(s2 / n - mean * mean, 0.f);
    const float rstd = rsqrtf(var + eps);
    if ((c % cpg) == 0) {
      save_mean[r * groups + c / cpg] = mean;
      save_rstd[r * groups + c / cpg] = rstd;
    }
    const float4 g = *reinterpret_cast<const float4*>(gamma + c);
    const float4 b = *reinterpret_cast<const float4*>(beta + c);
    for (int t = 0; t < T; ++t) {
      const size_t o = ((size_t)r * T + t) * C + c;
      const float4 v = *reinterpret_cast<const float4*>(x + o);
      float4 w;
      w.x = (v.x - mean) * rstd * g.x + b.x;
      w.y = (v.y - mean) * rstd * g.y + b.y;
      w.z = (v.z - mean) * rstd * g.z + b.z;
      w.w = (v.w - mean) * rstd * g.w + b.w;
      *reinterpret_cast<float4*>(y + o) = w;
    }
  }
}

__global__ __launch_bounds__(256) void groupnorm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ save_mean,
                                                            const float* __restrict__ save_rstd,
                                                            float* __restrict__ dx, float* __restrict__ dgp,
                                                            float* __restrict__ dbp, int T, int C, int groups) {
  const int r = blockIdx.x, tid = threadIdx.x;
  const int cpg = C / groups, lanes = cpg / 4;
  for (int c = tid * 4; c < C; c += 1024) {
    const float mean = save_mean[r * groups + c / cpg], rstd = save_rstd[r * groups + c / cpg];
    const float4 g = *reinterpret_cast<const float4*>(gamma + c);
    float4 dg = make_float4(0, 0, 0, 0), db = dg;
    float s1 = 0.f, s2 = 0.f;  // sum(g*dy), sum(g*dy*xhat) over the group
    for (int t = 0; t < T; ++t) {
      const size_t o = ((size_t)r * T + t) * C + c;
      const float4 v = *reinterpret_cast<const float4*>(x + o);
      const float4 d = *reinterpret_cast<const float4*>(dy + o);
      const float hx = (v.x - mean) * rstd, hy = (v.y - mean) * rstd, hz = (v.z - mean) * rstd,
                  hw = (v.w - mean) * rstd;
      dg.x += d.x * hx; dg.y += d.y * hy; dg.z += d.z * hz; dg.w += d.w * hw;
      db.x += d.x; db.y += d.y; db.z += d.z; db.w += d.w;
      s1 += (g.x * d.x + g.y * d.y) + (g.z * d.z + g.w * d.w);
      s2 += (g.x * d.x * hx + g.y * d.y * hy) + (g.z * d.z * hz + g.w * d.w * hw);
    }
    for (int o = 1; o < lanes; o <<= 1) {
      s1 += __shfl_xor(s1, o);
      s2 += __shfl_xor(s2, o);
    }
    *reinterpret_cast<float4*>(dgp + (size_t)r * C + c) = dg;
    *reinterpret_cast<float4*>(dbp + (size_t)r * C + c) = db;
    const float n = (float)(cpg * T);
    const float m1 = s1 / n, m2 = s2 / n;
    for (int t = 0; t < T; ++t) {
      const size_t o = ((size_t)r * T + t) * C + c;
      const float4 v = *reinterpret_cast<const float4*>(x + o);
      const float4 d = *reinterpret_cast<const float4*>(dy + o);
      float4 w;
      w.x = rstd * (g.x * d.x - m1 - (v.x - mean) * rstd * m2);
      w.y = rstd * (g.y * d.y - m1 - (v.y - mean) * rstd * m2);
      w.z = rstd * (g.z * d.z - m1 - (v.z - mean) * rstd * m2);
      w.w = rstd * (g.w * d.w - m1 - (v.w - mean) * rstd * m2);
      *reinterpret_cast<float4*>(dx + o) = w;
    }
  }
}

// out[c] = sum_rows x[row][c]  (fixed order -> deterministic)
// Column sums (bias gradient of a linear layer).  One workgroup per 32 columns: 8 row lanes x 32 column lanes, a row
// lane walks rows y, y + 8, ... with four independent partial sums (the former one-thread-per-column loop was a
// 1536-long dependent chain: 94 us on the PE / in_proj matrices of config 3), then a fixed-order LDS reduction.
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, int x_ld, float* __restrict__ out,
                                                     int rows, int cols) {
  __shared__ float red[8][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + tx;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (c < cols) {
    const float* p = x + c;
    int r = ty;
    for (; r + 24 < rows; r += 32) {
      s0 += p[(size_t)r * x_ld];
      s1 += p[(size_t)(r + 8) * x_ld];
      s2 += p[(size_t)(r + 16) * x_ld];
      s3 += p[(size_t)(r + 24) * x_ld];
    }
    for (; r < rows; r += 8) s0 += p[(size_t)r * x_ld];
  }
  red[ty][tx] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (ty == 0 && c < cols) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += red[k][tx];
    out[c] = s;
  }
}

// ---------------------------------------------------------------- MHA core, L_q = 1
// grid = r*heads waves (4 waves per workgroup).  lane owns 4 of the head's d=e/heads dims per
// 256-wide chunk (d = 256 -> exactly one float4 per lane).
template <int MAXT>
__global__ __launch_bounds__(256) void mha_q1_fwd_kernel(const float* __restrict__ q, const float* __restrict__ kv,
                                                         const float* __restrict__ drop, float* __restrict__ ctx,
                                                         float* __restrict__ probs, float* __restrict__ pdrop_out,
                                                         int R, int T, int E, int heads, float scale) {
  const int wid = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (wid >= R * heads) return;
  const int r = wid / heads, h = wid - r * heads;
  const int d = E / heads;
  float sc[MAXT];
#pragma unroll
  for (int t = 0; t < MAXT; ++t) sc[t] = 0.f;
  for (int j = lane * 4; j < d; j += 256) {
    const float4 qv = *reinterpret_cast<const float4*>(q + (size_t)r * E + h * d + j);
#pragma unroll
    for (int t = 0; t < MAXT; ++t)
      if (t < T) {
        const float4 kk = *reinterpret_cast<const float4*>(kv + ((size_t)r * T + t) * 2 * E + h * d + j);
        sc[t] += (qv.x * kk.x + qv.y * kk.y) + (qv.z * kk.z + qv.w * kk.w);
      }
  }
  float mx = -INFINITY;
#pragma unroll
  for (int t = 0; t < MAXT; ++t)
    if (t < T) {
      sc[t] = wave_sum(sc[t]) * scale;
      mx = fmaxf(mx, sc[t]);
    }
  float den = 0.f;
#pragma unroll
  for (int t = 0; t < MAXT; ++t)
    if (t < T) {
      sc[t] = expf(sc[t] - mx);
      den += sc[t];
    }
  const float inv = 1.f / den;
#pragma unroll
  for (int t = 0; t < MAXT; ++t)
    if (t < T) {
      const float p = sc[t] * inv;
      const float pd = drop ? p * drop[((size_t)r * heads + h) * T + t] : p;
      if (lane == 0) {
        probs[((size_t)r * heads + h) * T + t] = p;
        pdrop_out[((size_t)r * heads + h) * T + t] = pd;
      }
      sc[t] = pd;
    }
  for (int j = lane * 4; j < d; j += 256) {
    float4 acc = make_float4(0, 0, 0, 0);
#pragma unroll
    for (int t = 0; t < MAXT; ++t)
      if (t < T) {
        const float4 vv = *reinterpret_cast<const float4*>(kv + ((size_t)r * T + t) * 2 * E + E + h * d + j);
        acc.x = fmaf(sc[t], vv.x, acc.x);
        acc.y = fmaf(sc[t], vv.y, acc.y);
        acc.z = fmaf(sc[t], vv.z, acc.z);
        acc.w = fmaf(sc[t], vv.w, acc.w);
      }
    *reinterpret_cast<float4*>(ctx + (size_t)r * E + h * d + j) = acc;
  }
}

// avg_w[r][t] = mean_h pdrop[r][h][t]
__global__ void head_mean_kernel(const float* __restrict__ pd, float* __restrict__ avg, int R, int T, int heads) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= R * T) return;
  const int r = i / T, t = i - r * T;
  float s = 0.f;
  for (int h = 0; h < heads; ++h) s += pd[((size_t)r * heads + h) * T + t];
  avg[i] = s / (float)heads;
}

template <int MAXT>
__global__ __launch_bounds__(256) void mha_q1_bwd_kernel(const float* __restrict__ dctx,
                                                         const float* __restrict__ davg, const float* __restrict__ q,
                                                         const float* __restrict__ kv,
                                                         const float* __restrict__ probs,
                                                         const float* __restrict__ drop, float* __restrict__ dq,
                                                         float* __restrict__ dkv, int R, int T, int E, int heads,
                                                         float scale) {
  const int wid = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (wid >= R * heads) return;
  const int r = wid / heads, h = wid - r * heads;
  const int d = E / heads;
  float dp[MAXT];
#pragma unroll
  for (int t = 0; t < MAXT; ++t) dp[t] = 0.f;
  // d(pdrop)[t] = dctx_h . v[t]
  for (int j = lane * 4; j < d; j += 256) {
    const float4 dc = *reinterpret_cast<const float4*>(dctx + (size_t)r * E + h * d + j);
#pragma unroll
    for (int t = 0; t < MAXT; ++t)
      if (t < T) {
        const float4 vv = *reinterpret_cast<const float4*>(kv + ((size_t)r * T + t) * 2 * E + E + h * d + j);
        dp[t] += (dc.x * vv.x + dc.y * vv.y) + (dc.z * vv.z + dc.w * vv.w);
      }
  }
  float p[MAXT], pd[MAXT];
  float dot = 0.f;
#pragma unroll
  for (int t = 0; t < MAXT; ++t)
    if (t < T) {
      const size_t o = ((size_t)r * heads + h) * T + t;
      const float m = drop ? drop[o] : 1.f;
      p[t] = probs[o];
      pd[t] = p[t] * m;
      float g = wave_sum(dp[t]);
      if (davg) g += davg[(size_t)r * T + t] / (float)heads;
      dp[t] = g * m;  // gradient wrt the pre-dropout softmax output
      dot += p[t] * dp[t];
    } else {
      p[t] = pd[t] = 0.f;
    }
  float ds[MAXT];
#pragma unroll
  for (int t = 0; t < MAXT; ++t) ds[t] = (t < T) ? scale * p[t] * (dp[t] - dot) : 0.f;
  for (int j = lane * 4; j < d; j += 256) {
    const float4 qv = *reinterpret_cast<const float4*>(q + (size_t)r * E + h * d + j);
    const float4 dc = *reinterpret_cast<const float4*>(dctx + (size_t)r * E + h * d + j);
    float4 dqa = make_float4(0, 0, 0, 0);
#pragma unroll
    for (int t = 0; t < MAXT; ++t)
      if (t < T) {
        const size_t ko = ((size_t)r * T + t) * 2 * E + h * d + j;
        const float4 kk = *reinterpret_cast<const float4*>(kv + ko);
        dqa.x = fmaf(ds[t], kk.x, dqa.x);
        dqa.y = fmaf(ds[t], kk.y, dqa.y);
        dqa.z = fmaf(ds[t], kk.z, dqa.z);
        dqa.w = fmaf(ds[t], kk.w, dqa.w);
        *reinterpret_cast<float4*>(dkv + ko) = make_float4(ds[t] * qv.x, ds[t] * qv.y, ds[t] * qv.z, ds[t] * qv.w);
        *reinterpret_cast<float4*>(dkv + ko + E) = make_float4(pd[t] * dc.x, pd[t] * dc.y, pd[t] * dc.z, pd[t] * dc.w);
      }
    *reinterpret_cast<float4*>(dq + (size_t)r * E + h * d + j) = dqa;
  }
}

// ---------------------------------------------------------------- fixed attention, consensus, masks
__global__ __launch_bounds__(256) void weighted_sum_fwd_kernel(const float* __restrict__ feat,
                                                               const float* __restrict__ w, float* __restrict__ out,
                                                               int out_ld, int R, int T, int C) {
  const size_t total = (size_t)R * C;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int c = (int)(i % C);
    const size_t r = i / C;
    float s = 0.f;
    for (int t = 0; t < T; ++t) s = fmaf(feat[(r * T + t) * C + c], w[r * T + t], s);
    out[r * out_ld + c] = s;
  }
}
__global__ __launch_bounds__(256) void weighted_sum_bwd_kernel(const float* __restrict__ dout, int dout_ld,
                                                               const float* __restrict__ w,
                                                               float* __restrict__ dfeat, int R, int T, int C) {
  const size_t total = (size_t)R * T * C;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int c = (int)(i % C);
    const size_t rt = i / C, r = rt / T;
    dfeat[i] = dout[r * dout_ld + c] * w[rt];
  }
}
__global__ __launch_bounds__(256) void segment_mean_fwd_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                               int B, int N, int C) {
  const size_t total = (size_t)B * C;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int c = (int)(i % C);
    const size_t b = i / C;
    float s = 0.f;
    for (int n = 0; n < N; ++n) s += x[(b * N + n) * C + c];
    out[i] = s / (float)N;
  }
}
__global__ __launch_bounds__(256) void segment_mean_bwd_kernel(const float* __restrict__ dout, float* __restrict__ dx,
                                                               int B, int N, int C) {
  const size_t total = (size_t)B * N * C;
  const float k = 1.f / (float)N;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int c = (int)(i % C);
    const size_t b = i / C / N;
    dx[i] = dout[b * C + c] * k;
  }
}
__global__ __launch_bounds__(256) void mul_mask_kernel(const float* x, const float* __restrict__ m, float* y,
                                                       size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) y[i] = x[i] * m[i];
}
__global__ __launch_bounds__(256) void relu_mask_bwd_kernel(const float* dy, const float* __restrict__ y,
                                                            const float* __restrict__ m, float* dx, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    float g = y[i] > 0.f ? dy[i] : 0.f;
    if (m) g *= m[i];
    dx[i] = g;
  }
}

// dropout from a caller-drawn uniform tensor: mask = rnd >= p ? 1 / (1 - p) : 0 (torch's nn.Dropout rule, the draw itself
// stays torch's RNG), y = x * mask; the compare / cast / scale / multiply chain of four elementwise launches in one
__global__ __launch_bounds__(256) void dropout_fwd_kernel(const float* __restrict__ x, const float* __restrict__ rnd,
                                                          float p, float keep_scale, float* __restrict__ y,
                                                          float* __restrict__ mask, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float m = rnd[i] >= p ? keep_scale : 0.f;
    mask[i] = m;
    y[i] = x[i] * m;
  }
}

// Cross entropy of several classification heads that share one score matrix (reference model.py:272-279: one
// nn.CrossEntropyLoss per class key, summed): head h owns the columns [col0[h], col0[h] + ncls[h]) of scores (B, ld) and
// int64 labels label[h][b].  One workgroup per (head, sample): max / sum-exp by wavefront reduction, then
//   rowloss[h * B + b] = logsumexp - score[label],   dscores[b][c] = (softmax_c - [c == label]) / B   (columns outside every
// head are left untouched: the caller zero-fills).  The mean over b is a fixed-order sum in ce_heads_mean_kernel
// (deterministic).  A label of -100 is nn.CrossEntropyLoss's default ignore_index: the row gives no loss and no gradient
// and the mean divides by the number of the other rows; any other out-of-range label poisons the loss with NaN instead of
// reading out of bounds (torch raises a device assert there).
struct CeHeads {
  int n;
  int col0[4], ncls[4];
  const long long* label[4];
};
__global__ __launch_bounds__(256) void ce_heads_fwd_kernel(const float* __restrict__ scores, int ld, CeHeads hd, int B,
                                                           float* __restrict__ rowloss, float* __restrict__ dscores) {
  __shared__ float red[8];
  __shared__ int cnt[4];
  const int h = blockIdx.x / B, b = blockIdx.x - h * B;
  const int C = hd.ncls[h];
  const float* s = scores + (size_t)b * ld + hd.col0[h];
  float* d = dscores + (size_t)b * ld + hd.col0[h];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // nn.CrossEntropyLoss(ignore_index=-100, reduction="mean"): a row labelled -100 contributes neither loss nor gradient
  // and the mean runs over the OTHER rows -- every workgroup counts them itself (B int64 labels: a few hundred bytes)
  int valid = 0;
  for (int i = tid; i < B; i += 256) valid += hd.label[h][i] != -100 ? 1 : 0;
  for (int o = 32; o > 0; o >>= 1) valid += __shfl_xor(valid, o);
  float mx = -INFINITY;
  for (int c = tid; c < C; c += 256) mx = fmaxf(mx, s[c]);
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  if (lane == 0) {
    red[wave] = mx;
    cnt[wave] = valid;
  }
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  valid = (cnt[0] + cnt[1]) + (cnt[2] + cnt[3]);
  float se = 0.f;
  for (int c = tid; c < C; c += 256) se += expf(s[c] - mx);
  for (int o = 32; o > 0; o >>= 1) se += __shfl_xor(se, o);
  if (lane == 0) red[4 + wave] = se;
  __syncthreads();
  se = (red[4] + red[5]) + (red[6] + red[7]);
  const long long lab = hd.label[h][b];
  const bool ignored = lab == -100;
  const bool ok = lab >= 0 && lab < C;
  const float lse = mx + logf(se), inv = 1.f / se, invB = ignored ? 0.f : 1.f / (float)valid;
  for (int c = tid; c < C; c += 256)
    d[c] = ignored ? 0.f : (expf(s[c] - mx) * inv - ((long long)c == lab ? 1.f : 0.f)) * invB;
  if (tid == 0) rowloss[h * B + b] = ignored ? 0.f : (ok ? lse - s[lab] : NAN);   // any other out-of-range label: NaN
}
__global__ __launch_bounds__(64) void ce_heads_mean_kernel(const float* __restrict__ rowloss, CeHeads hd, int B, int nheads,
                                                           float* __restrict__ loss) {
  const int h = blockIdx.x;
  if (h >= nheads) return;
  double acc = 0.0;
  int valid = 0;
  for (int b = threadIdx.x; b < B; b += 64) {
    acc += (double)rowloss[h * B + b];
    valid += hd.label[h][b] != -100 ? 1 : 0;
  }
  for (int o = 32; o > 0; o >>= 1) {
    acc += __shfl_xor(acc, o);
    valid += __shfl_xor(valid, o);
  }
  if (threadIdx.x == 0) loss[h] = (float)(acc / (double)valid);   // no valid row: 0 / 0 = NaN, as torch
}
// dscores *= per-head upstream gradient (d total / d loss_h): the backward of the fused cross entropy
__global__ __launch_bounds__(256) void ce_heads_bwd_kernel(const float* __restrict__ dsc, int ld, CeHeads hd, int B,
                                                           const float* __restrict__ upstream, float* __restrict__ out) {
  const int h = blockIdx.x / B, b = blockIdx.x - h * B;
  const float g = upstream[h];
  const size_t o = (size_t)b * ld + hd.col0[h];
  for (int c = threadIdx.x; c < hd.ncls[h]; c += 256) out[o + c] = dsc[o + c] * g;
}

extern "C" {

int tbn_dropout_fwd(const float* x, const float* rnd, float p, float* y, float* mask, size_t count, void* stream) {
  TBN_REQUIRE(x && rnd && y && mask && p >= 0.f && p < 1.f, "dropout_fwd: bad argument (p = %g)", (double)p);
  TBN_KLAUNCH(dropout_fwd_kernel, dim3(ew_grid(count)), dim3(256), 0, (hipStream_t)stream, x, rnd, p, 1.f / (1.f - p), y,
              mask, count);
  TBN_CHECK_LAUNCH("dropout_fwd");
  return TBN_OK;
}

int tbn_ce_heads_fwd(const float* scores, int ld, int batch, int num_heads, const int* col0, const int* ncls,
                     const long long* const* labels, float* rowloss, float* loss, float* dscores, void* stream) {
  TBN_REQUIRE(scores && col0 && ncls && labels && rowloss && loss && dscores, "ce_heads_fwd: null argument");
  TBN_REQUIRE(num_heads >= 1 && num_heads <= 4 && batch >= 1, "ce_heads_fwd: 1..4 heads, batch >= 1");
  CeHeads hd;
  hd.n = num_heads;
  for (int h = 0; h < num_heads; ++h) {
    TBN_REQUIRE(col0[h] >= 0 && ncls[h] >= 1 && col0[h] + ncls[h] <= ld && labels[h] != nullptr,
                "ce_heads_fwd: head %d covers columns [%d, %d) of %d", h, col0[h], col0[h] + ncls[h], ld);
    hd.col0[h] = col0[h];
    hd.ncls[h] = ncls[h];
    hd.label[h] = labels[h];
  }
  TBN_KLAUNCH(ce_heads_fwd_kernel, dim3(num_heads * batch), dim3(256), 0, (hipStream_t)stream, scores, ld, hd, batch, rowloss,
              dscores);
  TBN_CHECK_LAUNCH("ce_heads_fwd");
  TBN_KLAUNCH(ce_heads_mean_kernel, dim3(num_heads), dim3(64), 0, (hipStream_t)stream, rowloss, hd, batch, num_heads, loss);
  TBN_CHECK_LAUNCH("ce_heads_mean");
  return TBN_OK;
}

int tbn_ce_heads_bwd(const float* dscores, int ld, int batch, int num_heads, const int* col0, const int* ncls,
                     const float* upstream, float* out, void* stream) {
  TBN_REQUIRE(dscores && col0 && ncls && upstream && out && num_heads >= 1 && num_heads <= 4 && batch >= 1,
              "ce_heads_bwd: bad argument");
  CeHeads hd;
  hd.n = num_heads;
  for (int h = 0; h < num_heads; ++h) {
    TBN_REQUIRE(col0[h] >= 0 && ncls[h] >= 1 && col0[h] + ncls[h] <= ld, "ce_heads_bwd: head %d out of range", h);
    hd.col0[h] = col0[h];
    hd.ncls[h] = ncls[h];
    hd.label[h] = nullptr;
  }
  TBN_KLAUNCH(ce_heads_bwd_kernel, dim3(num_heads * batch), dim3(256), 0, (hipStream_t)stream, dscores, ld, hd, batch,
              upstream, out);
  TBN_CHECK_LAUNCH("ce_heads_bwd");
  return TBN_OK;
}

int tbn_pe_concat_fwd(const float* feat, int feat_ld, const float* pe, float* out, int out_ld, int r, int t, int c,
                      int pe_dim, void* stream) {
  TBN_REQUIRE(out_ld >= c + pe_dim, "pe_concat: out_ld too small");
  TBN_KLAUNCH(pe_concat_kernel, dim3(ew_grid((size_t)r * t * out_ld)), dim3(256), 0, (hipStream_t)stream, feat,
                     feat_ld, pe, out, out_ld, r, t, c, pe_dim);
  TBN_CHECK_LAUNCH("pe_concat");
  return TBN_OK;
}

static int gn_ok(int c, int groups) {
  if (groups <= 0 || c % groups) return 0;
  const int cpg = c / groups;
  return cpg % 4 == 0 && cpg <= 256 && ((cpg / 4) & (cpg / 4 - 1)) == 0;
}

int tbn_groupnorm_fwd(const float* x, float* y, const float* gamma, const float* beta, float* save_mean,
                      float* save_rstd, int r, int t, int c, int groups, float eps, void* stream) {
  TBN_REQUIRE(gn_ok(c, groups), "groupnorm: channels/group must be 4*2^k (<=256)");
  TBN_KLAUNCH(groupnorm_fwd_kernel, dim3(r), dim3(256), 0, (hipStream_t)stream, x, y, gamma, beta, save_mean,
                     save_rstd, t, c, groups, eps);
  TBN_CHECK_LAUNCH("groupnorm_fwd");
  return TBN_OK;
}

int tbn_groupnorm_bwd(const float* dy, const float* x, const float* gamma, const float* save_mean,
                      const float* save_rstd, float* dx, float* dgamma_part, float* dbeta_part, int r, int t, int c,
                      int groups, void* stream) {
  TBN_REQUIRE(gn_ok(c, groups), "groupnorm: channels/group must be 4*2^k (<=256)");
  TBN_KLAUNCH(groupnorm_bwd_kernel, dim3(r), dim3(256), 0, (hipStream_t)stream, dy, x, gamma, save_mean,
                     save_rstd, dx, dgamma_part, dbeta_part, t, c, groups);
  TBN_CHECK_LAUNCH("groupnorm_bwd");
  return TBN_OK;
}

int tbn_colsum(const float* x, int x_ld, float* out, int rows, int cols, void* stream) {
  TBN_KLAUNCH(colsum_kernel, dim3(cdiv(cols, 32)), dim3(256), 0, (hipStream_t)stream, x, x_ld, out, rows, cols);
  TBN_CHECK_LAUNCH("colsum");
  return TBN_OK;
}

int tbn_mha_q1_fwd(const float* q, const float* kv, const float* drop_mask, float* ctx, float* probs, float* avg_w,
                   int r, int t, int e, int heads, float scale, void* stream) {
  TBN_REQUIRE(t >= 1 && t <= 32 && heads >= 1 && e % heads == 0 && (e / heads) % 4 == 0,
              "mha_q1: need 1<=T<=32 and head_dim %% 4 == 0");
  hipStream_t st = (hipStream_t)stream;
  // ctx doubles as nothing else; post-dropout probabilities are staged in avg_w's tail is not possible
  // (r*t floats only), so they go to `probs + r*heads*t` -- the probs buffer is 2*r*heads*t floats.
  float* pdrop = probs + (size_t)r * heads * t;
  const int blocks = cdiv(r * heads, 4);
  if (t <= 16)
    TBN_KLAUNCH((mha_q1_fwd_kernel<16>), dim3(blocks), dim3(256), 0, st, q, kv, drop_mask, ctx, probs, pdrop, r,
                       t, e, heads, scale);
  else
    TBN_KLAUNCH((mha_q1_fwd_kernel<32>), dim3(blocks), dim3(256), 0, st, q, kv, drop_mask, ctx, probs, pdrop, r,
                       t, e, heads, scale);
  TBN_CHECK_LAUNCH("mha_q1_fwd");
  TBN_KLAUNCH(head_mean_kernel, dim3(cdiv(r * t, 256)), dim3(256), 0, st, pdrop, avg_w, r, t, heads);
  TBN_CHECK_LAUNCH("head_mean");
  return TBN_OK;
}

int tbn_mha_q1_bwd(const float* dctx, const float* davg_w, const float* q, const float* kv, const float* probs,
                   const float* drop_mask, float* dq, float* dkv, int r, int t, int e, int heads, float scale,
                   void* stream) {
  TBN_REQUIRE(t >= 1 && t <= 32 && heads >= 1 && e % heads == 0 && (e / heads) % 4 == 0,
              "mha_q1: need 1<=T<=32 and head_dim %% 4 == 0");
  hipStream_t st = (hipStream_t)stream;
  const int blocks = cdiv(r * heads, 4);
  if (t <= 16)
    TBN_KLAUNCH((mha_q1_bwd_kernel<16>), dim3(blocks), dim3(256), 0, st, dctx, davg_w, q, kv, probs, drop_mask,
                       dq, dkv, r, t, e, heads, scale);
  else
    TBN_KLAUNCH((mha_q1_bwd_kernel<32>), dim3(blocks), dim3(256), 0, st, dctx, davg_w, q, kv, probs, drop_mask,
                       dq, dkv, r, t, e, heads, scale);
  TBN_CHECK_LAUNCH("mha_q1_bwd");
  return TBN_OK;
}

int tbn_weighted_sum_fwd(const float* feat, const float* w, float* out, int out_ld, int r, int t, int c, void* stream) {
  TBN_KLAUNCH(weighted_sum_fwd_kernel, dim3(ew_grid((size_t)r * c)), dim3(256), 0, (hipStream_t)stream, feat, w,
                     out, out_ld, r, t, c);
  TBN_CHECK_LAUNCH("weighted_sum_fwd");
  return TBN_OK;
}
int tbn_weighted_sum_bwd(const float* dout, int dout_ld, const float* w, float* dfeat, int r, int t, int c,
                         void* stream) {
  TBN_KLAUNCH(weighted_sum_bwd_kernel, dim3(ew_grid((size_t)r * t * c)), dim3(256), 0, (hipStream_t)stream,
                     dout, dout_ld, w, dfeat, r, t, c);
  TBN_CHECK_LAUNCH("weighted_sum_bwd");
  return TBN_OK;
}
int tbn_segment_mean_fwd(const float* x, float* out, int b, int n, int c, void* stream) {
  TBN_KLAUNCH(segment_mean_fwd_kernel, dim3(ew_grid((size_t)b * c)), dim3(256), 0, (hipStream_t)stream, x, out,
                     b, n, c);
  TBN_CHECK_LAUNCH("segment_mean_fwd");
  return TBN_OK;
}
int tbn_segment_mean_bwd(const float* dout, float* dx, int b, int n, int c, void* stream) {
  TBN_KLAUNCH(segment_mean_bwd_kernel, dim3(ew_grid((size_t)b * n * c)), dim3(256), 0, (hipStream_t)stream,
                     dout, dx, b, n, c);
  TBN_CHECK_LAUNCH("segment_mean_bwd");
  return TBN_OK;
}
int tbn_mul_mask(const float* x, const float* mask, float* y, size_t count, void* stream) {
  TBN_KLAUNCH(mul_mask_kernel, dim3(ew_grid(count)), dim3(256), 0, (hipStream_t)stream, x, mask, y, count);
  TBN_CHECK_LAUNCH("mul_mask");
  return TBN_OK;
}
int tbn_relu_mask_bwd(const float* dy, const float* y, const float* mask, float* dx, size_t count, void* stream) {
  TBN_KLAUNCH(relu_mask_bwd_kernel, dim3(ew_grid(count)), dim3(256), 0, (hipStream_t)stream, dy, y, mask, dx,
                     count);
  TBN_CHECK_LAUNCH("relu_mask_bwd");
  return TBN_OK;
}

}  // extern "C"
