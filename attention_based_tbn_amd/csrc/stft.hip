// Log-power STFT spectrogram on gfx950 (replaces the CPU librosa call of reference
// core/dataset/dataset.py:461-495: librosa.stft(n_fft=511, hop=120, win_length=240, "hann",
// center=True, pad_mode="constant") followed by log(|X|^2 + 1e-6)).
//
// Only 239 of the 511 window taps are non-zero (periodic Hann(240), centred), so a frame's 256
// bins are a dense 256 x 240 real DFT: X[k] = sum_j hann[j] * y[120 t - 120 + j] * e^{-2 pi i (j+135) k / 511}.
// That is a [bins x taps] x [taps x frames] contraction -> fp32 MFMA (32x32x2), with the window
// folded into a twiddle table (host fp64 -> fp32, 2 x 256 taps x 256 bins, L2 resident).
// One workgroup = 32 frames x 256 bins of one segment; frames are staged as an im2col tile in LDS
// ([32][257] -> conflict-free column reads); the accumulator layout puts frames on lanes so the
// (256, W) freq-major output rows are written 128 B contiguous.
#include <cmath>
#include <cstring>

#include "tbn_common.h"
#include "../../include/tbn_hip.h"

#define STFT_TAPS 256  // 240 real taps, zero padded
#define STFT_BINS 256

__global__ __launch_bounds__(256) void stft_logpower_kernel(const float* __restrict__ wave, int len, int W,
                                                            const float* __restrict__ tw, float* __restrict__ spec,
                                                            float eps) {
  __shared__ float fr[32 * 257];
  const int seg = blockIdx.y, t0 = blockIdx.x * 32;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const float* y = wave + (size_t)seg * len;
  for (int i = tid; i < 32 * 256; i += 256) {
    const int f = i >> 8, j = i & 255;
    const int idx = (t0 + f) * 120 - 120 + j;
    fr[f * 257 + j] = (j < 240 && idx >= 0 && idx < len) ? y[idx] : 0.f;
  }
  __syncthreads();
  const float* tc = tw;                           // [tap][bin] hann*cos
  const float* ts = tw + STFT_TAPS * STFT_BINS;   // [tap][bin] -hann*sin
  const int lrow = lane & 31, lhalf = lane >> 5;
  const int b0 = wv * 64;
  f32x16 re0, im0, re1, im1;
#pragma unroll
  for (int e = 0; e < 16; ++e) re0[e] = im0[e] = re1[e] = im1[e] = 0.f;
  for (int s = 0; s < 120; ++s) {
    const int tap = 2 * s + lhalf;
    const float b = fr[lrow * 257 + tap];
    const float c0 = tc[tap * STFT_BINS + b0 + lrow], s0 = ts[tap * STFT_BINS + b0 + lrow];
    const float c1 = tc[tap * STFT_BINS + b0 + 32 + lrow], s1 = ts[tap * STFT_BINS + b0 + 32 + lrow];
    re0 = __builtin_amdgcn_mfma_f32_32x32x2f32(c0, b, re0, 0, 0, 0);
    im0 = __builtin_amdgcn_mfma_f32_32x32x2f32(s0, b, im0, 0, 0, 0);
    re1 = __builtin_amdgcn_mfma_f32_32x32x2f32(c1, b, re1, 0, 0, 0);
    im1 = __builtin_amdgcn_mfma_f32_32x32x2f32(s1, b, im1, 0, 0, 0);
  }
  const int t = t0 + lrow;
  if (t < W) {
    float* o = spec + (size_t)seg * STFT_BINS * W + t;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = 8 * (e >> 2) + 4 * lhalf + (e & 3);
      o[(size_t)(b0 + row) * W] = logf(re0[e] * re0[e] + im0[e] * im0[e] + eps);
      o[(size_t)(b0 + 32 + row) * W] = logf(re1[e] * re1[e] + im1[e] * im1[e] + eps);
    }
  }
}

extern "C" {

size_t tbn_stft_twiddle_floats(void) { return (size_t)2 * STFT_TAPS * STFT_BINS; }

int tbn_stft_make_twiddle(float* host) {
  TBN_REQUIRE(host != nullptr, "stft_make_twiddle: null buffer");
  memset(host, 0, tbn_stft_twiddle_floats() * sizeof(float));
  const double pi = 3.14159265358979323846;
  for (int j = 0; j < 240; ++j) {
    const double h = 0.5 - 0.5 * cos(2.0 * pi * j / 240.0);
    for (int k = 0; k < STFT_BINS; ++k) {
      const long m = ((long)(j + 135) * k) % 511;  // exact phase reduction
      const double a = 2.0 * pi * (double)m / 511.0;
      host[j * STFT_BINS + k] = (float)(h * cos(a));
      host[STFT_TAPS * STFT_BINS + j * STFT_BINS + k] = (float)(-h * sin(a));
    }
  }
  return TBN_OK;
}

int tbn_stft_logpower(const float* wave, int nseg, int len, const float* twiddle, float* spec, float eps,
                      void* stream) {
  TBN_REQUIRE(wave && twiddle && spec && nseg > 0 && len > 0, "stft_logpower: bad argument");
  TBN_REQUIRE(nseg <= 65535, "stft_logpower: too many segments per call");
  const int W = 1 + (len - 1) / 120;
  hipLaunchKernelGGL(stft_logpower_kernel, dim3(cdiv(W, 32), nseg), dim3(256), 0, (hipStream_t)stream, wave, len, W,
                     twiddle, spec, eps);
  TBN_CHECK_LAUNCH("stft_logpower");
  return TBN_OK;
}

}  // extern "C"
