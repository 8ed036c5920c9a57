// How fast does an fp32-MFMA loop run as a function of the time since its kernel started?  Every wave stamps
// s_memrealtime (100 MHz) and s_memtime (shader clock) every 2 x 32 MFMAs; the host prints, per 5-us bin since kernel
// start, the shader clock and the MFMAs retired per SIMD per 64 cycles (1.0 = the pipe is full).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("hip error %d line %d\n", (int)e_, __LINE__); exit(1); } } while (0)
#define NS 64

struct Rec { long long rt[NS], ck[NS]; };

__global__ __launch_bounds__(1024) void k(const float* in, float* out, Rec* rec, int iters) {
  f32x16 acc[4];
  for (int a = 0; a < 4; ++a) for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
  float x[8], y[8];
  for (int i = 0; i < 8; ++i) { x[i] = in[(threadIdx.x * 16 + i) & 4095]; y[i] = in[(threadIdx.x * 16 + 8 + i + blockIdx.x) & 4095]; }
  Rec* q = rec + ((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  const bool w = (threadIdx.x & 63) == 0;
  for (int it = 0; it < iters; ++it) {
    if ((it & 1) == 0 && w && (it >> 1) < NS) { q->rt[it >> 1] = __builtin_amdgcn_s_memrealtime(); q->ck[it >> 1] = __builtin_readcyclecounter(); }
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int a = 0; a < 4; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(x[i], y[(i + a) & 7], acc[a], 0, 0, 0);
  }
  float r = 0; for (int a = 0; a < 4; ++a) for (int e = 0; e < 16; ++e) r += acc[a][e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

int main() {
  std::vector<float> h(4096);
  srand(1);
  for (auto& v : h) v = (rand() / (float)RAND_MAX - 0.5f) * 2.f;
  float *din, *d; Rec* drec;
  const int maxw = 1024 * 4;
  CK(hipMalloc(&din, 4096 * 4)); CK(hipMalloc(&d, (size_t)maxw * 64 * 4)); CK(hipMalloc(&drec, maxw * sizeof(Rec)));
  CK(hipMemcpy(din, h.data(), 4096 * 4, hipMemcpyHostToDevice));
  const int cfgs[][3] = {{256, 256, 126}, {768, 256, 60}, {768, 256, 126}, {256, 768, 60}};
  for (int idle_ms : {0, 20})
  for (auto& c : cfgs) {
    const int blocks = c[0], threads = c[1], iters = c[2], nw = blocks * threads / 64, ns = std::min(NS, iters / 2);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, din, d, drec, iters);
    CK(hipDeviceSynchronize());
    if (idle_ms) { struct timespec ts = {0, idle_ms * 1000000L}; nanosleep(&ts, nullptr); hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, din, d, drec, iters); CK(hipDeviceSynchronize()); }
    std::vector<Rec> r(nw);
    CK(hipMemcpy(r.data(), drec, nw * sizeof(Rec), hipMemcpyDeviceToHost));
    long long t0 = r[0].rt[0];
    for (auto& q : r) t0 = std::min(t0, q.rt[0]);
    // per 5-us bin: MFMAs retired (each interval of 64 MFMAs spread uniformly over its time span), clock of the intervals inside
    const int NB = 64; std::vector<double> mf(NB, 0.0), ckc(NB, 0.0), ckt(NB, 0.0);
    for (auto& q : r)
      for (int s = 0; s + 1 < ns; ++s) {
        const double a = (q.rt[s] - t0) * 0.01, b = (q.rt[s + 1] - t0) * 0.01;   // us
        if (b <= a) continue;
        for (int bin = (int)(a / 5); bin <= (int)(b / 5) && bin < NB; ++bin) {
          const double lo = std::max(a, bin * 5.0), hi = std::min(b, bin * 5.0 + 5.0);
          if (hi > lo) { mf[bin] += 64.0 * (hi - lo) / (b - a); ckc[bin] += (double)(q.ck[s + 1] - q.ck[s]) * (hi - lo) / (b - a); ckt[bin] += hi - lo; }
        }
      }
    printf("grid %4d x %4d threads, %3d x 32 MFMAs per wave, %s:\n   t(us) : pipe-fill | clock GHz\n", blocks, threads, iters, idle_ms ? "after 20 ms idle" : "back to back");
    for (int bin = 0; bin < NB; ++bin) {
      if (ckt[bin] <= 0) continue;
      // 1024 SIMDs, one MFMA = 64 cycles at the measured clock
      const double ghz = ckc[bin] / ckt[bin] * 1e-3;
      printf("   %3d-%3d: %5.2f | %5.3f\n", bin * 5, bin * 5 + 5, mf[bin] * 64.0 / (1024.0 * 5.0 * ghz * 1e3), ghz);
    }
  }
  return 0;
}
