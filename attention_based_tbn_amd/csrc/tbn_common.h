// Internal helpers shared by the HIP translation units of libtbn_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stddef.h>

#define TBN_OK 0
#define TBN_ERR_ARG (-1)
#define TBN_ERR_LAUNCH (-2)
#define TBN_ERR_UNSUPPORTED (-3)

// thread-local last-error string (tbn_last_error() in the C-ABI)
void tbn_set_error(const char* fmt, ...);

#define TBN_REQUIRE(cond, ...)                   \
  do {                                           \
    if (!(cond)) {                               \
      tbn_set_error(__VA_ARGS__);                \
      return TBN_ERR_ARG;                        \
    }                                            \
  } while (0)

#define TBN_CHECK_LAUNCH(what)                                                     \
  do {                                                                             \
    hipError_t e__ = hipGetLastError();                                            \
    if (e__ != hipSuccess) {                                                       \
      tbn_set_error("%s: launch failed: %s", what, hipGetErrorString(e__));        \
      return TBN_ERR_LAUNCH;                                                       \
    }                                                                              \
  } while (0)

// A/B experiment knobs.  The SHIPPED library reads no environment variable: tbn_env_int() is the constant `def` there
// (the compiler folds every `static const int knob = tbn_env_int(...)` away).  Only a -DTBN_EXPERIMENT=1 build
// (TBN_EXPERIMENT=1 python -m attention_based_tbn_amd.build, always a scripts/ab/ variant library selected with TBN_LIB)
// looks the variable up: validated ONCE -- a value outside [lo, hi] or not a number is refused (the default is used) and
// every knob that is set is announced on stderr (round-4 advisor, round-5 verdict item 7).
#ifndef TBN_EXPERIMENT
#define TBN_EXPERIMENT 0
#endif
#if TBN_EXPERIMENT
int tbn_env_int(const char* name, int def, int lo, int hi);
#else
static inline int tbn_env_int(const char*, int def, int, int) { return def; }
#endif

// Opt-in kernel TIMELINE (tbn_timeline_enable / tbn_timeline_dump in include/tbn_hip.h): every launch of the library gets a
// pair of events that ride on its dispatch packet (begin / end of the kernel itself), so that one un-traced multi-stream
// step can be laid out on a common clock -- rocprofv3's kernel trace makes the step host-bound (~40 us per intercepted
// launch) and the modality streams stop overlapping, which is exactly what the timeline is meant to show.
bool tbn_tl_on(hipStream_t st);   // recording AND `st` is not inside a stream capture
void tbn_tl_events(const char* name, hipStream_t st, hipEvent_t* start, hipEvent_t* stop);
#define TBN_KLAUNCH(kernel, grid, block, lds, st, ...)                                              \
  do {                                                                                              \
    if (tbn_tl_on(st)) {                                                                            \
      hipEvent_t tl_a__, tl_b__;                                                                    \
      tbn_tl_events(#kernel, st, &tl_a__, &tl_b__);                                                 \
      hipExtLaunchKernelGGL(kernel, grid, block, lds, st, tl_a__, tl_b__, 0, __VA_ARGS__);          \
    } else {                                                                                        \
      hipLaunchKernelGGL(kernel, grid, block, lds, st, __VA_ARGS__);                                \
    }                                                                                               \
  } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// exact unsigned division by a runtime constant for dividends < 2^31:
//   q = (m * mul) >> (31 + sh), sh = ceil(log2 d), mul = ceil(2^(31+sh) / d)  (fits 32 bits)
struct FastDiv {
  uint32_t mul, sh, d;
};
static inline FastDiv make_fastdiv(uint32_t d) {
  FastDiv f;
  f.d = d;
  uint32_t sh = 0;
  while ((1u << sh) < d) ++sh;
  f.sh = sh;
  uint64_t p = 1ull << (31 + sh);
  f.mul = (uint32_t)((p + d - 1) / d);
  return f;
}
__device__ __forceinline__ uint32_t fdiv(uint32_t m, FastDiv f) {
  return (uint32_t)(((uint64_t)m * f.mul) >> (31 + f.sh));
}

// flat element index -> (float4 channel quad g, x, y, n) of an (N, H, W, G) tensor with three exact magic-number
// divisions (~12 VALU instructions; the 64-bit `%` / `/` chains the elementwise kernels used before cost ~100 -- VALU
// time the fp32 MFMAs running beside these HBM-bound kernels do not get).  Requires N*H*W*G < 2^31.
struct PixDecode {
  FastDiv g, w, h;
};
static inline PixDecode make_pixdecode(int G, int W, int H) {
  PixDecode d;
  d.g = make_fastdiv((uint32_t)G);
  d.w = make_fastdiv((uint32_t)W);
  d.h = make_fastdiv((uint32_t)H);
  return d;
}
__device__ __forceinline__ void pix_decode(const PixDecode& d, uint32_t i, int& g, int& x, int& y, int& n, uint32_t& pix) {
  pix = fdiv(i, d.g);
  g = (int)(i - pix * d.g.d);
  const uint32_t row = fdiv(pix, d.w);
  x = (int)(pix - row * d.w.d);
  const uint32_t nn = fdiv(row, d.h);
  y = (int)(row - nn * d.h.d);
  n = (int)nn;
}

// up to 3 destination column ranges of a conv / bn output (concat-slice writes)
struct Seg {
  float* ptr;     // base of column `col_begin`
  int ld;         // floats between consecutive rows (pixels)
  int col_begin;  // first logical output column written through this segment
};
struct CSeg {
  const float* ptr;
  int ld;
  int col_begin;
};

#ifdef __HIPCC__
// Cache policy of the once-read streams of the HBM-bound passes (-DTBN_BN_NT=<bits> for A/B builds; shipped: 11).  A non-temporal
// load asks L2 not to keep the line: the 31 GB per step these passes move then displace less of what the conv GEMMs of the other
// modality streams share through L2 (weights, the row panels sibling workgroups re-read).  bit 0: the forward apply's y (not read
// again before the backward pass); bit 1: the backward apply's dz and y (their last use); bit 3: y in the stem's pooled
// BN-backward apply, the split-K slabs in their reduce, weights / gradients / momenta in the optimiser; bit 2: the STORES of
// z / dy -- their consumers follow at once, measured slower, off.  Same values either way (a cache hint): bit-identical results.
// Round 6, same-box alternations: config 4 -0.06 ... -0.09 ms per step (6 of 7), config 3 -0.2 ms, config 2 -0.05 ms
// (profiles/r06_ab_nontemporal_loads.txt).
#ifndef TBN_BN_NT
#define TBN_BN_NT 11
#endif
typedef float tbn_f32x4_nt __attribute__((ext_vector_type(4)));
template <bool NT>
__device__ __forceinline__ float4 tbn_ld4(const float* p) {
  if (NT) {
    const tbn_f32x4_nt v = __builtin_nontemporal_load(reinterpret_cast<const tbn_f32x4_nt*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
  }
  return *reinterpret_cast<const float4*>(p);
}
template <bool NT>
__device__ __forceinline__ void tbn_st4(float* p, const float4 o) {
  if (NT) {
    tbn_f32x4_nt v = {o.x, o.y, o.z, o.w};
    __builtin_nontemporal_store(v, reinterpret_cast<tbn_f32x4_nt*>(p));
  } else {
    *reinterpret_cast<float4*>(p) = o;
  }
}
#endif
