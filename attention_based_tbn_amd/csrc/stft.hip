// Log-power STFT spectrogram on gfx950 (replaces the CPU librosa call of reference
// core/dataset/dataset.py:461-495: librosa.stft(n_fft=511, hop=120, win_length=240, "hann",
// center=True, pad_mode="constant") followed by log(|X|^2 + 1e-6)).
//
// Only 239 of the 511 window taps are non-zero (periodic Hann(240), centred), so a frame's 256
// bins are a dense 256 x 240 real DFT: X[k] = sum_j hann[j] * y[120 t - 120 + j] * e^{-2 pi i (j+135) k / 511}.
// That is a [bins x taps] x [taps x frames] contraction -> fp32 MFMA (32x32x2), 2 * 2 * 256 * 240 FLOP per frame.
//
// Workgroup = 4 waves = 64 frames of one segment x 128 bins (wave = 32 bins; blockIdx.z picks the half of the bins); re and
// im of 2 frame tiles = 4 accumulators per wave.  Config 4's 96 segments x 4 frame blocks x 2 = 768 workgroups = 3 per CU.
// (An 8-wave form that stages the frames once for all 256 bins is 1.5 % faster when its grid happens to fill the CUs evenly
// and 25 % slower when it does not; a persistent form that overlaps the stores of one unit with the MFMAs of the next
// measured the same as this one -- both removed, round 3.)
//   * frames (B operand): hop = 120, 240 taps -> consecutive frames overlap by half.  The RAW samples of the 64
//     frames (65 hop rows of 120) are staged ONCE per workgroup in LDS with rows padded to 124 floats: frame f, tap j
//     lives at (f + j / 120) * 124 + j % 120, no im2col copy, and a lane's four consecutive taps are one conflict-free
//     ds_read_b128 (lane stride 496 B -> 16 distinct 16-B slots per 16-lane group).  The 65 rows are one contiguous
//     span of the waveform: staged with 16-B buffer loads (1950 of them per workgroup, 8 per thread) whose hardware
//     range check supplies the zero padding in front of the first and behind the last sample;
//   * twiddles (A operand): the window is folded into the table, stored [cos | -sin][tap / 8][bin][tap % 8]: the
//     fragment of 8 taps x 32 bins is ONE fully coalesced 1-KB load per wave (the table is 480 KB, shared by every
//     workgroup: L2 resident), prefetched two 8-tap groups ahead (stft_groups below); each load feeds 8 MFMAs;
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
__device__ void stft_buffer_store_f32(float v, stft_i32x4 srsrc, int voffset, int soffset, int aux) __asm(
    "llvm.amdgcn.raw.buffer.store.f32");

// One tap group (8 taps) = 16 MFMAs of a wave.  The loop runs D groups per trip on D-slot register rings, so no slot
// is copied: group kg multiplies slot kg % D while the twiddle fragments of groups kg + 1 .. kg + D - 1 and the frame
// taps of group kg + 1 are in flight into the other slots.  The twiddle address is a scalar base (advanced per group, wrapping to
// group 0 behind the last one: a persistent workgroup's ring runs on into its next unit) plus one per-lane offset, and
// there is no branch in the trip: with a conditional prefetch the compiler's wait-count pass fell back to vmcnt(0) right
// behind the loads it had just issued (the L2 latency sat in front of every group).
struct StftAcc {
  f32x16 re0, im0, re1, im1;
};
template <int D>
struct StftRing {
  float4 c[D], s[D];   // twiddle fragments (cos | -sin) of D consecutive tap groups
};

__device__ __forceinline__ float4 stft_tw_load(const float* tw, unsigned k_byte, unsigned lane_byte) {
  return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(tw) + k_byte + lane_byte);
}

// groups [KB, KE) of the K loop (KB, KE multiples of D); fr_off = LDS float index of (frame lrow, tap 8 * KB + 4 * lhalf);
// fa / fb [KB % 2] hold the frame taps of group KB on entry
template <int D, int KB, int KE>
__device__ __forceinline__ void stft_groups(const float* __restrict__ frc, int fr_off, const float* __restrict__ tw,
                                            unsigned lane_byte, StftRing<D>& tr, float4 (&fa)[D], float4 (&fb)[D], StftAcc& a) {
  static_assert(KB % D == 0 && KE % D == 0 && STFT_KG % D == 0, "ring depth must divide the group ranges");
  constexpr unsigned kGroupBytes = STFT_BINS * 8 * 4, kPartBytes = STFT_KG * kGroupBytes;
#pragma unroll 1
  for (int kg = KB; kg < KE; kg += D) {
#pragma unroll
    for (int j = 0; j < D; ++j) {
      const int k = kg + j;
      const unsigned kn = (unsigned)(k + D - 1 < STFT_KG ? k + D - 1 : k + D - 1 - STFT_KG) * kGroupBytes;
      tr.c[(j + D - 1) % D] = stft_tw_load(tw, kn, lane_byte);
      tr.s[(j + D - 1) % D] = stft_tw_load(tw, kn + kPartBytes, lane_byte);
      // taps of the next group: + 8 floats, + 12 where tap 120 opens the next hop row; the last group re-reads itself
      fr_off += k + 1 == STFT_KG ? 0 : (k + 1 == 15 ? 12 : 8);
      fa[(j + 1) % D] = *reinterpret_cast<const float4*>(&frc[fr_off]);
      fb[(j + 1) % D] = *reinterpret_cast<const float4*>(&frc[fr_off + 32 * STFT_PITCH]);
      __builtin_amdgcn_sched_barrier(0);
      const float4 c = tr.c[j], s = tr.s[j], f0 = fa[j], f1 = fb[j];
#define STFT_STEP(q)                                                            \
      a.re0 = __builtin_amdgcn_mfma_f32_32x32x2f32(c.q, f0.q, a.re0, 0, 0, 0);   \
      a.im0 = __builtin_amdgcn_mfma_f32_32x32x2f32(s.q, f0.q, a.im0, 0, 0, 0);   \
      a.re1 = __builtin_amdgcn_mfma_f32_32x32x2f32(c.q, f1.q, a.re1, 0, 0, 0);   \
      a.im1 = __builtin_amdgcn_mfma_f32_32x32x2f32(s.q, f1.q, a.im1, 0, 0, 0);
      STFT_STEP(x) STFT_STEP(y) STFT_STEP(z) STFT_STEP(w)
#undef STFT_STEP
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

template <int D>
__device__ __forceinline__ void stft_ring_start(const float* __restrict__ tw, unsigned lane_byte, StftRing<D>& tr) {
  constexpr unsigned kGroupBytes = STFT_BINS * 8 * 4, kPartBytes = STFT_KG * kGroupBytes;
#pragma unroll
  for (int d = 0; d + 1 < D; ++d) {
    tr.c[d] = stft_tw_load(tw, d * kGroupBytes, lane_byte);
    tr.s[d] = stft_tw_load(tw, d * kGroupBytes + kPartBytes, lane_byte);
  }
}

// log(|X|^2 + eps) of a wave's 32 bins x 64 frames -> (256, W) freq-major rows of the segment, 128 B contiguous per store;
// |X|^2 + eps >= 1e-6 is a normal number: v_log_f32 (log2, 1 ulp) * ln 2.  Raw-buffer stores: frames >= W get an offset
// outside the segment's 256 x W floats and are dropped by the range check (no branch per store, 32-bit offsets).
__device__ __forceinline__ void stft_store(float* __restrict__ o, int W, int b0, int t0, int lrow, int lhalf, float eps,
                                           const StftAcc& a) {
  const float ln2 = 0.6931471805599453f;
  union {
    stft_i32x4 v;
    struct {
      void* p;
      unsigned range, cfg;
    } d;
  } u;
  u.d.p = o;
  u.d.range = (unsigned)(STFT_BINS * W) * 4u;
  u.d.cfg = 0x00020000u;
  stft_i32x4 rs;
  rs.x = __builtin_amdgcn_readfirstlane(u.v.x);
  rs.y = __builtin_amdgcn_readfirstlane(u.v.y);
  rs.z = __builtin_amdgcn_readfirstlane(u.v.z);
  rs.w = __builtin_amdgcn_readfirstlane(u.v.w);
  const int ta = t0 + lrow, tb = ta + 32;
  const int row = (b0 + 4 * lhalf) * W;
  const int oa = ta < W ? (row + ta) * 4 : (int)0x80000000, ob = tb < W ? (row + tb) * 4 : (int)0x80000000;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int soff = (8 * (e >> 2) + (e & 3)) * W * 4;   // the bin row of accumulator element e (wave-uniform)
    stft_buffer_store_f32(__builtin_amdgcn_logf(fmaf(a.re0[e], a.re0[e], fmaf(a.im0[e], a.im0[e], eps))) * ln2, rs, oa + soff, 0, 0);
    stft_buffer_store_f32(__builtin_amdgcn_logf(fmaf(a.re1[e], a.re1[e], fmaf(a.im1[e], a.im1[e], eps))) * ln2, rs, ob + soff, 0, 0);
  }
}

// a segment's samples as a raw buffer: offsets in front of sample 0 (negative -> huge unsigned) and behind sample len - 1
// are out of range, per dword -> zeros (librosa's centre padding, reference dataset.py:487-489)
__device__ __forceinline__ stft_i32x4 stft_segment_buffer(const float* wave, int seg, int len) {
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
  return rs;
}

__global__ __launch_bounds__(256) void stft_logpower_kernel(const float* __restrict__ wave, int len, int W,
                                                            const float* __restrict__ tw, float* __restrict__ spec, float eps) {
  __shared__ __attribute__((aligned(16))) float fr[(STFT_FR + 1) * STFT_PITCH];
  const int seg = blockIdx.y, t0 = blockIdx.x * STFT_FR;
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lrow = lane & 31, lhalf = lane >> 5;
  const int b0 = blockIdx.z * 128 + wv * 32;
  const unsigned lane_byte = (unsigned)((b0 + lrow) * 2 + lhalf) * 16u;   // [part][kg][b0 + lrow][4 * lhalf]
  StftRing<3> tr;
  stft_ring_start(tw, lane_byte, tr);
  {
    const stft_i32x4 rs = stft_segment_buffer(wave, seg, len);
    const int g0 = (t0 - 1) * 120;     // first staged sample (a multiple of 4: a 16-B group never straddles sample 0)
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int i4 = tid + 256 * q;    // 16-B group of the 65 x 120 span (30 groups per hop row)
      if (i4 < (STFT_FR + 1) * 30) {
        const int r = (i4 * 2185) >> 16, c4 = i4 - r * 30;   // i4 / 30 for i4 < 2048
        const f32x4 v = stft_buffer_load_f32x4(rs, (g0 + i4 * 4) * 4, 0, 0);
        *reinterpret_cast<float4*>(&fr[r * STFT_PITCH + c4 * 4]) = make_float4(v.x, v.y, v.z, v.w);
      }
    }
  }
  StftAcc a;
#pragma unroll
  for (int e = 0; e < 16; ++e) a.re0[e] = a.im0[e] = a.re1[e] = a.im1[e] = 0.f;
  __syncthreads();
  const int off = lrow * STFT_PITCH + 4 * lhalf;
  float4 fa[3], fb[3];
  fa[0] = *reinterpret_cast<const float4*>(&fr[off]);
  fb[0] = *reinterpret_cast<const float4*>(&fr[off + 32 * STFT_PITCH]);
  stft_groups<3, 0, STFT_KG>(fr, off, tw, lane_byte, tr, fa, fb, a);
  stft_store(spec + (size_t)seg * STFT_BINS * W, W, b0, t0, lrow, lhalf, eps, a);
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
  const int W = 1 + (len - 1) / 120;
  // 32-bit byte offsets inside a segment's samples and inside its 256 x W spectrogram
  TBN_REQUIRE(nseg <= 65535 && (size_t)len * 4 < (1ull << 31) && (size_t)STFT_BINS * W * 4 < (1ull << 31),
              "stft_logpower: too many segments / too long a waveform per call");
  TBN_KLAUNCH(stft_logpower_kernel, dim3(cdiv(W, STFT_FR), nseg, 2), dim3(256), 0, (hipStream_t)stream, wave, len, W,
                     twiddle, spec, eps);
  TBN_CHECK_LAUNCH("stft_logpower");
  return TBN_OK;
}

}  // extern "C"
