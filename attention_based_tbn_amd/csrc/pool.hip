// Pooling, layout and stem-weight packing kernels (NHWC fp32, 16 B per lane, HBM-bound).
//
// Reference ops (core/models/bn_inception_audio.py):
//   MaxPool2d(3, stride 2, ceil_mode=True)          pool1/pool2, inception_3c/4e pass-through (:21-23,32-34,155-157,327-329)
//   MaxPool2d(3, stride 1, pad 1)                   inception_5b_pool (:394-396)
//   AvgPool2d(3, stride 1, pad 1, count_include_pad) inception_{3a,3b,4a-4d,5a}_pool (:86-88 ...)
//   F.avg_pool2d over (H,1) or (H,W)                 BNInception.logits (core/models/bn_inception.py:16-35)
// Max-pool forward also stores the in-window argmax (uint8, first maximum in scan order like
// ATen) so the backward is a deterministic gather (no float atomics).
#include "tbn_common.h"
#include "tbn_kernels.h"
#include "tbn_pool_dev.h"

static inline int ew_grid(size_t items) {
  size_t g = (items + 255) / 256;
  return (int)(g > 8192 ? 8192 : (g < 1 ? 1 : g));
}

// Stencil kernels re-read their neighbours' rows: workgroups b, b+8, ... share an XCD (and its L2), so the
// bijective remap hands every XCD a CONTIGUOUS run of workgroup ids -- a row is then fetched into one L2, not
// into all eight (measured: 3x the algorithmic HBM reads without it).
__device__ __forceinline__ unsigned xcd_block_id() {
  const unsigned nb = gridDim.x, bid = blockIdx.x;
  const unsigned q8 = nb >> 3, r8 = nb & 7, xcd = bid & 7, idx = bid >> 3;
  return (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
}

__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const float* __restrict__ in, int in_ld,
                                                          float* __restrict__ out, int out_ld,
                                                          uint8_t* __restrict__ argmax, int N, int H, int W, int C,
                                                          int OH, int OW, int stride, int pad, PixDecode dec) {
  const int G = C >> 2;
  const uint32_t total = (uint32_t)N * OH * OW * G;
  for (uint32_t i = xcd_block_id() * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
    int g, ox, oy, n;
    uint32_t opix;
    pix_decode(dec, i, g, ox, oy, n, opix);
    const int y0 = oy * stride - pad, x0 = ox * stride - pad;
    float4 best = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    int bx = 0, by = 0, bz = 0, bw = 0;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const int iy = y0 + r, ix = x0 + s;
        if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) {
          const float4 v = *reinterpret_cast<const float4*>(in + ((size_t)(n * H + iy) * W + ix) * in_ld + g * 4);
          const int k = r * 3 + s;
          if (v.x > best.x || v.x != v.x) { best.x = v.x; bx = k; }
          if (v.y > best.y || v.y != v.y) { best.y = v.y; by = k; }
          if (v.z > best.z || v.z != v.z) { best.z = v.z; bz = k; }
          if (v.w > best.w || v.w != v.w) { best.w = v.w; bw = k; }
        }
      }
    *reinterpret_cast<float4*>(out + (size_t)opix * out_ld + g * 4) = best;
    if (argmax != nullptr)
      *reinterpret_cast<uint32_t*>(argmax + (size_t)opix * C + g * 4) =
          (uint32_t)bx | ((uint32_t)by << 8) | ((uint32_t)bz << 16) | ((uint32_t)bw << 24);
  }
}

int tbn_launch_maxpool_fwd(const float* in, int in_ld, float* out, int out_ld, uint8_t* argmax, int N, int H, int W,
                           int C, int OH, int OW, int stride, int pad, hipStream_t st) {
  TBN_REQUIRE(C % 4 == 0 && in_ld % 4 == 0 && out_ld % 4 == 0, "maxpool: C / pitches must be multiples of 4");
  TBN_REQUIRE((size_t)N * H * W * (C / 4) < (1ull << 31), "maxpool: too many elements per call");
  TBN_KLAUNCH(maxpool_fwd_kernel, dim3(ew_grid((size_t)N * OH * OW * C / 4)), dim3(256), 0, st, in, in_ld, out,
                     out_ld, argmax, N, H, W, C, OH, OW, stride, pad, make_pixdecode(C / 4, OW, OH));
  TBN_CHECK_LAUNCH("maxpool_fwd");
  return TBN_OK;
}

// gather form: each input element sums dout of the (<= 4 or 9) windows whose argmax it is
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ dout, int dout_ld,
                                                          const uint8_t* __restrict__ argmax,
                                                          float* __restrict__ din, int din_ld, int N, int H, int W,
                                                          int C, int OH, int OW, int stride, int pad, int accumulate,
                                                          PixDecode dec) {
  const int G = C >> 2;
  const uint32_t total = (uint32_t)N * H * W * G;
  for (uint32_t i = xcd_block_id() * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
    int g, ix, iy, n;
    uint32_t ipix;
    pix_decode(dec, i, g, ix, iy, n, ipix);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (stride == 2 && pad == 0) {
      // the pass-through pools of inception_3c / 4e: branch-free form of the loops below (window i >> 1 with tap
      // i & 1 and, for even i, window (i >> 1) - 1 with tap 2); all loads are issued together, same summation order
      const int oyA = iy >> 1, ryA = iy & 1, oxA = ix >> 1, rxA = ix & 1;
      const bool vy[2] = {oyA < OH, (ryA == 0) && (oyA >= 1)}, vx[2] = {oxA < OW, (rxA == 0) && (oxA >= 1)};
      const int oy[2] = {vy[0] ? oyA : 0, vy[1] ? oyA - 1 : 0}, ry[2] = {ryA, 2};
      const int ox[2] = {vx[0] ? oxA : 0, vx[1] ? oxA - 1 : 0}, rx[2] = {rxA, 2};
      uint32_t am[4];
      float4 d[4];
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const size_t opix = (size_t)(n * OH + oy[a]) * OW + ox[b];
          am[a * 2 + b] = *reinterpret_cast<const uint32_t*>(argmax + opix * C + g * 4);
          d[a * 2 + b] = *reinterpret_cast<const float4*>(dout + opix * dout_ld + g * 4);
        }
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const uint32_t k = (uint32_t)(ry[a] * 3 + rx[b]), m = am[a * 2 + b];
          const bool v = vy[a] && vx[b];
          if (v && (m & 0xff) == k) acc.x += d[a * 2 + b].x;
          if (v && ((m >> 8) & 0xff) == k) acc.y += d[a * 2 + b].y;
          if (v && ((m >> 16) & 0xff) == k) acc.z += d[a * 2 + b].z;
          if (v && (m >> 24) == k) acc.w += d[a * 2 + b].w;
        }
    } else {
    // windows oy with oy*stride - pad <= iy <= oy*stride - pad + 2
    const int oy_hi = min(OH - 1, (iy + pad) / stride);
    const int ox_hi = min(OW - 1, (ix + pad) / stride);
    for (int oy = oy_hi; oy >= 0; --oy) {
      const int r = iy - (oy * stride - pad);
      if (r > 2) break;
      for (int ox = ox_hi; ox >= 0; --ox) {
        const int s = ix - (ox * stride - pad);
        if (s > 2) break;
        const uint32_t k = (uint32_t)(r * 3 + s);
        const size_t opix = (size_t)(n * OH + oy) * OW + ox;
        const uint32_t am = *reinterpret_cast<const uint32_t*>(argmax + opix * C + g * 4);
        const float4 d = *reinterpret_cast<const float4*>(dout + opix * dout_ld + g * 4);
        if ((am & 0xff) == k) acc.x += d.x;
        if (((am >> 8) & 0xff) == k) acc.y += d.y;
        if (((am >> 16) & 0xff) == k) acc.z += d.z;
        if ((am >> 24) == k) acc.w += d.w;
      }
    }
    }
    float4* o = reinterpret_cast<float4*>(din + (size_t)ipix * din_ld + g * 4);
    if (accumulate) {
      const float4 prev = *o;
      acc.x += prev.x; acc.y += prev.y; acc.z += prev.z; acc.w += prev.w;
    }
    *o = acc;
  }
}

// 3x3 / stride 2 / pad 0 (pool1, pool2 when not fused, the pass-through pools of inception_3c / 4e): a thread owns a
// 2x2 input block, whose four pixels share the same four windows (tbn_pool_dev.h)
__global__ __launch_bounds__(256) void maxpool_bwd2x2_kernel(PoolBlk pb, float* __restrict__ din, int din_ld, int NQ,
                                                             int C, int accumulate, FastDiv divg) {
  const int G = C >> 2;
  const uint32_t total = (uint32_t)NQ * G;
  for (uint32_t i = xcd_block_id() * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
    const uint32_t q = fdiv(i, divg);
    const int c = (int)(i - q * (uint32_t)G) * 4;
    const uint32_t row = fdiv(q, pb.div_bw);
    const int bx = (int)q - (int)row * pb.BW;
    const uint32_t n = fdiv(row, pb.div_bh);
    const int by = (int)row - (int)n * pb.BH;
    float4 g[4];
    pooled_grad_2x2(pb, (int)n, by, bx, c, g);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int iy = 2 * by + (k >> 1), ix = 2 * bx + (k & 1);
      if (iy < pb.H && ix < pb.W) {
        float4* o = reinterpret_cast<float4*>(din + ((size_t)((int)n * pb.H + iy) * pb.W + ix) * din_ld + c);
        float4 acc = g[k];
        if (accumulate) {
          const float4 prev = *o;
          acc.x += prev.x; acc.y += prev.y; acc.z += prev.z; acc.w += prev.w;
        }
        *o = acc;
      }
    }
  }
}

int tbn_launch_maxpool_bwd(const float* dout, int dout_ld, const uint8_t* argmax, float* din, int din_ld, int N,
                           int H, int W, int C, int OH, int OW, int stride, int pad, int accumulate, hipStream_t st) {
  TBN_REQUIRE(C % 4 == 0 && din_ld % 4 == 0 && dout_ld % 4 == 0, "maxpool_bwd: C / pitches must be multiples of 4");
  TBN_REQUIRE((size_t)N * H * W * (C / 4) < (1ull << 31), "maxpool_bwd: too many elements per call");
  if (stride == 2 && pad == 0) {
    const int NQ = N * ((H + 1) / 2) * ((W + 1) / 2);
    TBN_KLAUNCH(maxpool_bwd2x2_kernel, dim3(ew_grid((size_t)NQ * C / 4)), dim3(256), 0, st,
                       make_poolblk(dout, dout_ld, argmax, H, W, OH, OW, C), din, din_ld, NQ, C, accumulate,
                       make_fastdiv((uint32_t)(C / 4)));
    TBN_CHECK_LAUNCH("maxpool_bwd2x2");
    return TBN_OK;
  }
  TBN_KLAUNCH(maxpool_bwd_kernel, dim3(ew_grid((size_t)N * H * W * C / 4)), dim3(256), 0, st, dout, dout_ld,
                     argmax, din, din_ld, N, H, W, C, OH, OW, stride, pad, accumulate, make_pixdecode(C / 4, W, H));
  TBN_CHECK_LAUNCH("maxpool_bwd");
  return TBN_OK;
}

// 3x3 / stride 1 / pad 1 / count_include_pad average.  Self-adjoint, so it is its own backward.
__global__ __launch_bounds__(256) void avgpool3_kernel(const float* __restrict__ in, int in_ld,
                                                       float* __restrict__ out, int out_ld, int N, int H, int W, int C,
                                                       int accumulate, PixDecode dec) {
  const int G = C >> 2;
  const uint32_t total = (uint32_t)N * H * W * G;
  for (uint32_t i = xcd_block_id() * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
    int g, x, y, n;
    uint32_t opix;
    pix_decode(dec, i, g, x, y, n, opix);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int r = -1; r <= 1; ++r)
#pragma unroll
      for (int s = -1; s <= 1; ++s) {
        const int iy = y + r, ix = x + s;
        if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) {
          const float4 v = *reinterpret_cast<const float4*>(in + ((size_t)(n * H + iy) * W + ix) * in_ld + g * 4);
          acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
      }
    const float k = 1.f / 9.f;
    acc.x *= k; acc.y *= k; acc.z *= k; acc.w *= k;
    float4* o = reinterpret_cast<float4*>(out + (size_t)opix * out_ld + g * 4);
    if (accumulate) {
      const float4 prev = *o;
      acc.x += prev.x; acc.y += prev.y; acc.z += prev.z; acc.w += prev.w;
    }
    *o = acc;
  }
}

int tbn_launch_avgpool3_fwd(const float* in, int in_ld, float* out, int out_ld, int N, int H, int W, int C,
                            int accumulate, hipStream_t st) {
  TBN_REQUIRE(C % 4 == 0 && in_ld % 4 == 0 && out_ld % 4 == 0, "avgpool: C / pitches must be multiples of 4");
  TBN_REQUIRE((size_t)N * H * W * (C / 4) < (1ull << 31), "avgpool: too many elements per call");
  TBN_KLAUNCH(avgpool3_kernel, dim3(ew_grid((size_t)N * H * W * C / 4)), dim3(256), 0, st, in, in_ld, out,
                     out_ld, N, H, W, C, accumulate, make_pixdecode(C / 4, W, H));
  TBN_CHECK_LAUNCH("avgpool3");
  return TBN_OK;
}

// mean over (H,W) [freq_only=0 -> out (N,C)] or over H only [freq_only=1 -> out (N,W,C)].
// One workgroup per output row (n [, x]) and 256-channel slice: 256 threads = 64 float4 channel lanes x 4 pixel lanes
// (all channel lanes when C < 256), the pixel lanes' partial sums are combined through LDS in fixed order.  (These two kernels sit alone between the end of a
// backbone's forward and the heads / at the start of its backward: the former one-thread-per-output form with 64-bit
// index divisions took 50 us for a 19 MB tensor.)
__global__ __launch_bounds__(256) void spatial_mean_fwd_kernel(const float* __restrict__ in, int in_ld,
                                                               float* __restrict__ out, int out_ld, int N, int H,
                                                               int W, int C, int freq_only) {
  __shared__ float4 red[256];
  const int G = C >> 2;                       // float4 lanes per pixel
  const int GL = G < 64 ? G : 64;             // channel lanes of this workgroup
  const int PL = 256 / GL;                    // pixel lanes
  const int gl = threadIdx.x % GL, pl = threadIdx.x / GL;
  const int g = blockIdx.y * GL + gl;
  const int q = blockIdx.x;                   // output row
  const int OW = freq_only ? W : 1;
  const int n = q / OW, x = q - n * OW;
  const int npix = freq_only ? H : H * W;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (pl < PL && g < G)
    for (int i = pl; i < npix; i += PL) {
      const int pix = freq_only ? (n * H + i) * W + x : n * npix + i;
      const float4 v = *reinterpret_cast<const float4*>(in + (size_t)pix * in_ld + g * 4);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
  red[threadIdx.x] = acc;
  __syncthreads();
  if (pl == 0 && g < G) {
    for (int k = 1; k < PL; ++k) {
      const float4 v = red[k * GL + gl];
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    const float k = 1.f / (float)npix;
    acc.x *= k; acc.y *= k; acc.z *= k; acc.w *= k;
    *reinterpret_cast<float4*>(out + (size_t)q * out_ld + g * 4) = acc;
  }
}

int tbn_launch_spatial_mean_fwd(const float* in, int in_ld, float* out, int out_ld, int N, int H, int W, int C,
                                int freq_only, hipStream_t st) {
  TBN_REQUIRE(C % 4 == 0 && C <= 1024 && in_ld % 4 == 0 && out_ld % 4 == 0,
              "spatial_mean: C / pitches must be multiples of 4, C <= 1024");
  const int G = C / 4, GL = G < 64 ? G : 64;
  TBN_KLAUNCH(spatial_mean_fwd_kernel, dim3(N * (freq_only ? W : 1), cdiv(G, GL)), dim3(256), 0, st, in, in_ld, out,
                     out_ld, N, H, W, C, freq_only);
  TBN_CHECK_LAUNCH("spatial_mean_fwd");
  return TBN_OK;
}

__global__ __launch_bounds__(256) void spatial_mean_bwd_kernel(const float* __restrict__ dout, int dout_ld,
                                                               float* __restrict__ din, int din_ld, int N, int H,
                                                               int W, int C, int freq_only) {
  const int G = C >> 2;
  const int total = N * H * W * G;            // < 2^31 (plan_create bounds the pixel count)
  const float k = 1.f / (float)(freq_only ? H : H * W);
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int pix = i / G, g = i - pix * G;
    const int row = pix / W, x = pix - row * W;
    const int n = row / H;
    const int q = freq_only ? n * W + x : n;
    float4 v = *reinterpret_cast<const float4*>(dout + (size_t)q * dout_ld + g * 4);
    v.x *= k; v.y *= k; v.z *= k; v.w *= k;
    *reinterpret_cast<float4*>(din + (size_t)pix * din_ld + g * 4) = v;
  }
}

int tbn_launch_spatial_mean_bwd(const float* dout, int dout_ld, float* din, int din_ld, int N, int H, int W, int C,
                                int freq_only, hipStream_t st) {
  TBN_REQUIRE(C % 4 == 0 && din_ld % 4 == 0 && dout_ld % 4 == 0, "spatial_mean_bwd: bad C / pitches");
  TBN_KLAUNCH(spatial_mean_bwd_kernel, dim3(ew_grid((size_t)N * H * W * C / 4)), dim3(256), 0, st, dout,
                     dout_ld, din, din_ld, N, H, W, C, freq_only);
  TBN_CHECK_LAUNCH("spatial_mean_bwd");
  return TBN_OK;
}

// NCHW is the reference tensor layout (model.py:211-213)
// ---------------------------------------------------------------- the 7x7 / stride 2 / pad 3 stem as a space-to-depth conv
// A 7x7 / stride-2 conv on C input channels is exactly a 4x4 / stride-1 conv on the 2x2 space-to-depth image
// (4*C channels = pixel parities (y&1, x&1) x C, half the height / width): x[2oy-3+r] with r+1 = 2a+py reads s2d row
// oy-2+a, parity py, so the taps are a, b in 0..3 with "pad 2" on the top / left and 1 on the bottom / right
// (a = py = 0 has no source tap: zero weight).  A filter row is then 4 s2d pixels x 4C channels = 16C CONTIGUOUS floats
// and K = 4 rows x 16C = 64C, of which 49C carry weights: 64 / 192 / 640 multiplied columns for audio / RGB / flow
// instead of the 224 / 224 / 672 of channel-padded NHWC filter rows (7 rows x 32-float chunks).
// The image is written WITH its zero border ([N][OH+3][OW+3][4C], OH = ceil(H/2)), so the GEMM kernels read it without
// any bounds logic (conv_igemm.hip ROWMODE); odd H / W just leave the missing parity zero.
template <int CT>   // CT > 0: compile-time channel count (float4 stores); 0: runtime C, scalar stores
__global__ __launch_bounds__(256) void nchw_to_s2d_pad_kernel(const float* __restrict__ in, float* __restrict__ out, int N,
                                                              int Crt, int H, int W, int HP, int WP) {
  const int C = CT > 0 ? CT : Crt;
  const size_t total = (size_t)N * HP * WP, hw = (size_t)H * W;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int X = (int)(i % WP);
    const size_t t = i / WP;
    const int Y = (int)(t % HP), n = (int)(t / HP);
    const int y0 = 2 * (Y - 2), x0 = 2 * (X - 2);
    const float* src = in + (size_t)n * C * hw;
    bool ok[4];
    size_t off[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int y = y0 + (q >> 1), x = x0 + (q & 1);
      ok[q] = (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
      off[q] = ok[q] ? (size_t)y * W + x : 0;
    }
    float* o = out + i * (size_t)(4 * C);
    if (CT > 0) {
      float v[4 * (CT > 0 ? CT : 1)];
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int c = 0; c < CT; ++c) v[q * CT + c] = ok[q] ? src[(size_t)c * hw + off[q]] : 0.f;
#pragma unroll
      for (int c = 0; c < CT; ++c)
        reinterpret_cast<float4*>(o)[c] = make_float4(v[4 * c], v[4 * c + 1], v[4 * c + 2], v[4 * c + 3]);
    } else {
      for (int q = 0; q < 4; ++q)
        for (int c = 0; c < C; ++c) o[q * C + c] = ok[q] ? src[(size_t)c * hw + off[q]] : 0.f;
    }
  }
}

int tbn_launch_nchw_to_s2d_pad(const float* in, float* out, int N, int C, int H, int W, hipStream_t st) {
  TBN_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0, "nchw_to_s2d_pad: empty input");
  const int HP = (H + 1) / 2 + 3, WP = (W + 1) / 2 + 3;
  const dim3 grid(ew_grid((size_t)N * HP * WP));
  switch (C) {
    case 1: TBN_KLAUNCH(nchw_to_s2d_pad_kernel<1>, grid, dim3(256), 0, st, in, out, N, C, H, W, HP, WP); break;
    case 3: TBN_KLAUNCH(nchw_to_s2d_pad_kernel<3>, grid, dim3(256), 0, st, in, out, N, C, H, W, HP, WP); break;
    case 10: TBN_KLAUNCH(nchw_to_s2d_pad_kernel<10>, grid, dim3(256), 0, st, in, out, N, C, H, W, HP, WP); break;
    default: TBN_KLAUNCH(nchw_to_s2d_pad_kernel<0>, grid, dim3(256), 0, st, in, out, N, C, H, W, HP, WP); break;
  }
  TBN_CHECK_LAUNCH("nchw_to_s2d_pad");
  return TBN_OK;
}

// weights [Cout][7][7][C] -> [Cout][K = 64C]: k = (a*4 + b)*4C + (py*2 + px)*C + c holds w[2a+py-1][2b+px-1][c]
// (zero where that tap lies outside 0..6)
__global__ void pack_stem_weight_s2d_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cout, int C) {
  const int K = 64 * C, total = Cout * K;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int co = i / K, k = i - co * K;
    const int ab = k / (4 * C), cq = k - ab * 4 * C, a = ab >> 2, b = ab & 3;
    const int par = cq / C, c = cq - par * C, py = par >> 1, px = par & 1;
    const int r = 2 * a + py - 1, s2 = 2 * b + px - 1;
    wp[i] = (r >= 0 && r < 7 && s2 >= 0 && s2 < 7) ? w[((co * 7 + r) * 7 + s2) * C + c] : 0.f;
  }
}
__global__ void unpack_stem_wgrad_s2d_kernel(const float* __restrict__ dwp, float* __restrict__ dw, int Cout, int C) {
  const int total = Cout * 49 * C;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int c = i % C, s2 = (i / C) % 7, r = (i / (7 * C)) % 7, co = i / (49 * C);
    const int a = (r + 1) >> 1, py = (r + 1) & 1, b = (s2 + 1) >> 1, px = (s2 + 1) & 1;
    dw[i] = dwp[(size_t)co * 64 * C + (a * 4 + b) * 4 * C + (py * 2 + px) * C + c];
  }
}
// ---------------------------------------------------------------- the 7x7 / stride 2 stem on a bordered NHWC image ("row runs")
// For many input channels the space-to-depth form above multiplies 64*C K columns for 49*C taps x channels.  On a
// zero-bordered NHWC image [N][HP][WP][C] (3 border pixels on the top / left) a filter ROW of output pixel (oy, ox) is
// the 7*C CONTIGUOUS floats starting at pixel (2 oy + r, 2 ox): the K of a pixel = 7 such runs, each rounded up to a
// multiple of 4 floats (RL; the extra floats are the next pixel's data and meet zero weights), packed back to back and
// padded to a multiple of 32 with a run "7" that also meets zero weights: K = 512 instead of 640 for the 10-channel flow
// stem (96 % useful columns instead of 77 %).  For 3 / 1 channels the 32-float K granularity gives 192 / 64 either way,
// so RGB / audio keep the space-to-depth form (its 16-B aligned loads).  The image is HP = 2 (OH - 1) + 8 rows,
// WP = 2 (OW - 1) + 8 columns: every run a launch reads lies inside it.
template <int CT>   // CT > 0: compile-time channel count; 0: runtime
__global__ __launch_bounds__(256) void nchw_to_nhwc_pad_kernel(const float* __restrict__ in, float* __restrict__ out, int N,
                                                               int Crt, int H, int W, int HP, int WP) {
  const int C = CT > 0 ? CT : Crt;
  const size_t total = (size_t)N * HP * WP, hw = (size_t)H * W;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int X = (int)(i % WP);
    const size_t t = i / WP;
    const int Y = (int)(t % HP), n = (int)(t / HP);
    const int y = Y - 3, x = X - 3;
    const bool ok = (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
    const float* src = in + (size_t)n * C * hw + (ok ? (size_t)y * W + x : 0);
    float* o = out + i * (size_t)C;
    if (CT > 0 && CT % 2 == 0) {
      float v[CT > 0 ? CT : 1];
#pragma unroll
      for (int c = 0; c < CT; ++c) v[c] = ok ? src[(size_t)c * hw] : 0.f;
#pragma unroll
      for (int c = 0; c < CT; c += 2) *reinterpret_cast<float2*>(o + c) = make_float2(v[c], v[c + 1]);   // pixel pitch 8-B aligned
    } else {
      for (int c = 0; c < C; ++c) o[c] = ok ? src[(size_t)c * hw] : 0.f;
    }
  }
}

int tbn_launch_nchw_to_nhwc_pad(const float* in, float* out, int N, int C, int H, int W, int HP, int WP, hipStream_t st) {
  TBN_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0 && HP >= H + 3 && WP >= W + 3, "nchw_to_nhwc_pad: bad extents");
  const dim3 grid(ew_grid((size_t)N * HP * WP));
  if (C == 10)
    TBN_KLAUNCH(nchw_to_nhwc_pad_kernel<10>, grid, dim3(256), 0, st, in, out, N, C, H, W, HP, WP);
  else
    TBN_KLAUNCH(nchw_to_nhwc_pad_kernel<0>, grid, dim3(256), 0, st, in, out, N, C, H, W, HP, WP);
  TBN_CHECK_LAUNCH("nchw_to_nhwc_pad");
  return TBN_OK;
}

// weights [Cout][7][7][C] -> [Cout][K]: k = r * RL + s * C + c holds w[r][s][c]; zero in the run padding and beyond 7 * RL
__global__ void pack_stem_weight_rows_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cout, int C, int RL, int K) {
  const int total = Cout * K;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int co = i / K, k = i - co * K;
    const int r = k / RL, q = k - r * RL;
    wp[i] = (r < 7 && q < 7 * C) ? w[((size_t)(co * 7 + r) * 7) * C + q] : 0.f;
  }
}
// packed weight gradient [Cout][7 * RL] -> [Cout][7][7][C]
__global__ void unpack_stem_wgrad_rows_kernel(const float* __restrict__ dwp, float* __restrict__ dw, int Cout, int C, int RL) {
  const int total = Cout * 49 * C;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int q = i % (7 * C), r = (i / (7 * C)) % 7, co = i / (49 * C);
    dw[i] = dwp[(size_t)co * 7 * RL + r * RL + q];
  }
}
int tbn_launch_pack_stem_weight_rows(const float* w, float* wp, int Cout, int C, int RL, int K, hipStream_t st) {
  TBN_KLAUNCH(pack_stem_weight_rows_kernel, dim3(cdiv(Cout * K, 256)), dim3(256), 0, st, w, wp, Cout, C, RL, K);
  TBN_CHECK_LAUNCH("pack_stem_weight_rows");
  return TBN_OK;
}
int tbn_launch_unpack_stem_wgrad_rows(const float* dwp, float* dw, int Cout, int C, int RL, hipStream_t st) {
  TBN_KLAUNCH(unpack_stem_wgrad_rows_kernel, dim3(cdiv(Cout * 49 * C, 256)), dim3(256), 0, st, dwp, dw, Cout, C, RL);
  TBN_CHECK_LAUNCH("unpack_stem_wgrad_rows");
  return TBN_OK;
}

int tbn_launch_pack_stem_weight_s2d(const float* w, float* wp, int Cout, int C, hipStream_t st) {
  TBN_KLAUNCH(pack_stem_weight_s2d_kernel, dim3(cdiv(Cout * 64 * C, 256)), dim3(256), 0, st, w, wp, Cout, C);
  TBN_CHECK_LAUNCH("pack_stem_weight_s2d");
  return TBN_OK;
}
int tbn_launch_unpack_stem_wgrad_s2d(const float* dwp, float* dw, int Cout, int C, hipStream_t st) {
  TBN_KLAUNCH(unpack_stem_wgrad_s2d_kernel, dim3(cdiv(Cout * 49 * C, 256)), dim3(256), 0, st, dwp, dw, Cout, C);
  TBN_CHECK_LAUNCH("unpack_stem_wgrad_s2d");
  return TBN_OK;
}
