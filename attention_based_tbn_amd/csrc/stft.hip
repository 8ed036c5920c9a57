// Log-power STFT spectrogram on gfx950 (replaces the CPU librosa call of reference
// core/dataset/dataset.py:461-495: librosa.stft(n_fft=511, hop=120, win_length=240, "hann",
// center=True, pad_mode="constant") followed by log(|X|^2 + 1e-6)).
//
// Only 239 of the 511 window taps are non-zero (periodic Hann(240), centred), so a frame's 256
// bins are a dense 256 x 240 real DFT: X[k] = sum_j hann[j] * y[120 t - 120 + j] * e^{-2 pi i (j+135) k / 511}.
// That is a [bins x taps] x [taps x frames] contraction -> fp32 MFMA (32x32x2), 2 * 2 * 256 * 240 FLOP per frame.
//
// Workgroup = 64 frames of one segment x ALL 256 bins (8 waves x 32 bins) or x 128 bins (4 waves, two workgroups per
// frame block); re and im of 2 frame tiles = 4 accumulators per wave.  The 8-wave form stages the frames once per
// block, but a launch is only (frame blocks x segments) workgroups: config 4's 96 segments x 4 blocks = 384 leave the 256
// CUs with 1 or 2 workgroups each (measured: 81 us against 72 us for the 768 balanced 4-wave workgroups), so the
// launcher takes the 8-wave form only when its grid fills the CUs evenly (e.g. 192 segments: 149 -> 122 us).
//   * frames (B operand): hop = 120, 240 taps -> consecutive frames overlap by half.  The RAW samples of the 64
//     frames (65 hop rows of 120) are staged ONCE per workgroup in LDS with rows padded to 124 floats: frame f, tap j
//     lives at (f + j / 120) * 124 + j % 120, no im2col copy, and a lane's four consecutive taps are one conflict-free
//     ds_read_b128 (lane stride 496 B -> 16 distinct 16-B slots per 16-lane group).  The 65 rows are one contiguous
//     span of the waveform: staged with 16-B buffer loads (1950 of them per workgroup, 4 or 8 per thread) whose hardware
//     range check supplies the zero padding in front of the first and behind the last sample;
//   * twiddles (A operand): the window is folded into the table, stored [cos | -sin][tap / 8][bin][tap % 8]: the
//     fragment of 8 taps x 32 bins is ONE fully coalesced 1-KB load per wave (the table is 480 KB, shared by every
//     workgroup: L2 resident), prefetched two 8-tap groups ahead; each load feeds 8 MFMAs;
//   * k is permuted identically for A and B (lane half h holds taps 4h..4h+3 of the group) so one 16-B fragment
//     feeds four MFMAs; accumulators put frames on lanes -> the (256, W) freq-major rows are written 128 B contiguous;
//   * log(|X|^2 + eps) with the hardware logarithm (v_log_f32, 1 ulp): 3 VALU instructions per output instead of the
//     ~20 of logf's software path (32 outputs per lane: the epilogue was 8 % of the kernel).
#include <cmath>
#include <cstring>

#include "tbn_common.h"
#include "../../include/tbn_hip.h"

#define STFT_TAPS 240
#define STFT_KG 30     // groups of 8 taps
#define STFT_BINS 256
#define STFT_FR 64     // frames per workgroup
#define STFT_PITCH 124 // LDS floats per hop row of 120 samples

typedef int stft_i32x4 __attribute__((ext_vector_type(4)));
__device__ f32x4 stft_buffer_load_f32x4(stft_i32x4 srsrc, int voffset, int soffset, int aux) __asm(
    "llvm.amdgcn.raw.buffer.load.v4f32");

template <int NW>   // waves per workgroup: 8 (all 256 bins) or 4 (128 bins, blockIdx.z picks the half)
__global__ __launch_bounds__(64 * NW) void stft_logpower_kernel(const float* __restrict__ wave, int len, int W,
                                                                const float* __restrict__ tw, float* __restrict__ spec,
                                                                float eps) {
  __shared__ __attribute__((aligned(16))) float fr[(STFT_FR + 1) * STFT_PITCH];
  const int seg = blockIdx.y, t0 = blockIdx.x * STFT_FR;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int lrow = lane & 31, lhalf = lane >> 5;
  const int b0 = (NW == 8 ? 0 : blockIdx.z * 128) + wv * 32;
  {
    // this segment's samples as a raw buffer: offsets in front of sample 0 (negative -> huge unsigned) and behind sample
    // len - 1 are out of range, per dword -> zeros (librosa's centre padding, reference dataset.py:487-489)
    union {
      stft_i32x4 v;
      struct {
        const void* p;
        unsigned range, cfg;
      } d;
    } u;
    u.d.p = wave + (size_t)seg * len;
    u.d.range = (unsigned)len * 4u;
    u.d.cfg = 0x00020000u;
    stft_i32x4 rs;
    rs.x = __builtin_amdgcn_readfirstlane(u.v.x);
    rs.y = __builtin_amdgcn_readfirstlane(u.v.y);
    rs.z = __builtin_amdgcn_readfirstlane(u.v.z);
    rs.w = __builtin_amdgcn_readfirstlane(u.v.w);
    const int g0 = (t0 - 1) * 120;     // first staged sample (a multiple of 4: a 16-B group never straddles sample 0)
#pragma unroll
    for (int q = 0; q < 2048 / (64 * NW); ++q) {
      const int i4 = tid + 64 * NW * q;    // 16-B group of the 65 x 120 span (30 groups per hop row)
      if (i4 < (STFT_FR + 1) * 30) {
        const int r = (i4 * 2185) >> 16, c4 = i4 - r * 30;   // i4 / 30 for i4 < 2048
        const f32x4 v = stft_buffer_load_f32x4(rs, (g0 + i4 * 4) * 4, 0, 0);
        *reinterpret_cast<float4*>(&fr[r * STFT_PITCH + c4 * 4]) = make_float4(v.x, v.y, v.z, v.w);
      }
    }
  }
  // twiddle fragment of tap group kg: 16 B at [part][kg][b0 + lrow][4 * lhalf]
  const float4* twc = reinterpret_cast<const float4*>(tw) + ((size_t)(b0 + lrow) * 2 + lhalf);
  const size_t part4 = (size_t)STFT_KG * STFT_BINS * 2;   // float4 per table
  f32x16 re0, im0, re1, im1;
#pragma unroll
  for (int e = 0; e < 16; ++e) re0[e] = im0[e] = re1[e] = im1[e] = 0.f;
  // twiddle fragments are prefetched TWO tap groups ahead: one group is 16 MFMAs (~1.3 us with three waves per SIMD),
  // about the L2 latency under load -- a distance of one group left the loads exposed
  float4 ac[2], as[2];
  ac[0] = twc[0];
  as[0] = twc[part4];
  ac[1] = twc[(size_t)STFT_BINS * 2];
  as[1] = twc[part4 + (size_t)STFT_BINS * 2];
  __syncthreads();
#pragma unroll 2
  for (int kg = 0; kg < STFT_KG; ++kg) {
    const float4 c = ac[kg & 1], s = as[kg & 1];
    if (kg + 2 < STFT_KG) {
      ac[kg & 1] = twc[(size_t)(kg + 2) * STFT_BINS * 2];
      as[kg & 1] = twc[part4 + (size_t)(kg + 2) * STFT_BINS * 2];
    }
    const int tap0 = kg * 8 + 4 * lhalf;
    const int off = (lrow + (tap0 >= 120 ? 1 : 0)) * STFT_PITCH + (tap0 >= 120 ? tap0 - 120 : tap0);
    const float4 f0 = *reinterpret_cast<const float4*>(&fr[off]);
    const float4 f1 = *reinterpret_cast<const float4*>(&fr[off + 32 * STFT_PITCH]);
#define STFT_STEP(q)                                                         \
    re0 = __builtin_amdgcn_mfma_f32_32x32x2f32(c.q, f0.q, re0, 0, 0, 0);      \
    im0 = __builtin_amdgcn_mfma_f32_32x32x2f32(s.q, f0.q, im0, 0, 0, 0);      \
    re1 = __builtin_amdgcn_mfma_f32_32x32x2f32(c.q, f1.q, re1, 0, 0, 0);      \
    im1 = __builtin_amdgcn_mfma_f32_32x32x2f32(s.q, f1.q, im1, 0, 0, 0);
    STFT_STEP(x) STFT_STEP(y) STFT_STEP(z) STFT_STEP(w)
#undef STFT_STEP
  }
  float* o = spec + (size_t)seg * STFT_BINS * W;
  const float ln2 = 0.6931471805599453f;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int bin = b0 + 8 * (e >> 2) + 4 * lhalf + (e & 3);
    const int ta = t0 + lrow, tb = t0 + 32 + lrow;
    // |X|^2 + eps >= 1e-6 is a normal number: v_log_f32 (log2, 1 ulp) * ln 2
    if (ta < W) o[(size_t)bin * W + ta] = __builtin_amdgcn_logf(fmaf(re0[e], re0[e], fmaf(im0[e], im0[e], eps))) * ln2;
    if (tb < W) o[(size_t)bin * W + tb] = __builtin_amdgcn_logf(fmaf(re1[e], re1[e], fmaf(im1[e], im1[e], eps))) * ln2;
  }
}

extern "C" {

size_t tbn_stft_twiddle_floats(void) { return (size_t)2 * STFT_KG * STFT_BINS * 8; }

int tbn_stft_make_twiddle(float* host) {
  TBN_REQUIRE(host != nullptr, "stft_make_twiddle: null buffer");
  const double pi = 3.14159265358979323846;
  const size_t part = (size_t)STFT_KG * STFT_BINS * 8;
  for (int j = 0; j < STFT_TAPS; ++j) {
    const double h = 0.5 - 0.5 * cos(2.0 * pi * j / 240.0);
    for (int k = 0; k < STFT_BINS; ++k) {
      const long m = ((long)(j + 135) * k) % 511;  // exact phase reduction
      const double a = 2.0 * pi * (double)m / 511.0;
      const size_t idx = ((size_t)(j / 8) * STFT_BINS + k) * 8 + (j % 8);
      host[idx] = (float)(h * cos(a));
      host[part + idx] = (float)(-h * sin(a));
    }
  }
  return TBN_OK;
}

int tbn_stft_logpower(const float* wave, int nseg, int len, const float* twiddle, float* spec, float eps,
                      void* stream) {
  TBN_REQUIRE(wave && twiddle && spec && nseg > 0 && len > 0, "stft_logpower: bad argument");
  TBN_REQUIRE(nseg <= 65535 && (size_t)len * 4 < (1ull << 31), "stft_logpower: too many segments / too long a waveform per call");
  const int W = 1 + (len - 1) / 120;
  // the 8-wave form when (frame blocks x segments) workgroups load the 256 CUs evenly (>= 87 % of whole rounds)
  const int blocks = cdiv(W, STFT_FR) * nseg, rounds = cdiv(blocks, 256);
  if (blocks * 8 >= rounds * 256 * 7)
    hipLaunchKernelGGL(stft_logpower_kernel<8>, dim3(cdiv(W, STFT_FR), nseg), dim3(512), 0, (hipStream_t)stream, wave, len,
                       W, twiddle, spec, eps);
  else
    hipLaunchKernelGGL(stft_logpower_kernel<4>, dim3(cdiv(W, STFT_FR), nseg, 2), dim3(256), 0, (hipStream_t)stream, wave,
                       len, W, twiddle, spec, eps);
  TBN_CHECK_LAUNCH("stft_logpower");
  return TBN_OK;
}

}  // extern "C"
