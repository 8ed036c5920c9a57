// Where do the waves of a launch land, and when?  The MFMA loop of mfma_short.hip; every wave records its HW_ID
// (XCC / SE / CU / SIMD) and s_memrealtime at start and end.  Host: waves per CU and per SIMD (min / max / histogram),
// spread of the start times, spread of the per-wave durations, for several grid shapes at equal work per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("hip error %d line %d\n", (int)e_, __LINE__); exit(1); } } while (0)

struct Rec { unsigned hw, xcc; long long t0, t1; };

__global__ __launch_bounds__(1024) void k(const float* in, float* out, Rec* rec, int iters) {
  const long long t0 = __builtin_amdgcn_s_memrealtime();
  f32x16 acc[4];
  for (int a = 0; a < 4; ++a) for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
  float x[8], y[8];
  for (int i = 0; i < 8; ++i) { x[i] = in[(threadIdx.x * 16 + i) & 4095]; y[i] = in[(threadIdx.x * 16 + 8 + i + blockIdx.x) & 4095]; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int a = 0; a < 4; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(x[i], y[(i + a) & 7], acc[a], 0, 0, 0);
  }
  float r = 0; for (int a = 0; a < 4; ++a) for (int e = 0; e < 16; ++e) r += acc[a][e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
  const long long t1 = __builtin_amdgcn_s_memrealtime();
  if ((threadIdx.x & 63) == 0) {
    Rec q;
    q.hw = __builtin_amdgcn_s_getreg(63492);    // HW_REG_HW_ID, 32 bits
    q.xcc = __builtin_amdgcn_s_getreg(63508);   // HW_REG_XCC_ID
    q.t0 = t0; q.t1 = t1;
    rec[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = q;
  }
}

int main() {
  std::vector<float> h(4096);
  srand(1);
  for (auto& v : h) v = (rand() / (float)RAND_MAX - 0.5f) * 2.f;
  float *din, *d; Rec* drec;
  const int maxw = 4096 * 16;
  CK(hipMalloc(&din, 4096 * 4)); CK(hipMalloc(&d, (size_t)maxw * 64 * 4)); CK(hipMalloc(&drec, maxw * sizeof(Rec)));
  CK(hipMemcpy(din, h.data(), 4096 * 4, hipMemcpyHostToDevice));
  const int cfgs[][3] = {{768, 256, 0}, {768, 256, 50000}, {256, 768, 0}, {1024, 256, 0}, {512, 512, 0}, {1536, 256, 0}, {2048, 256, 30000}};
  for (auto& c : cfgs) {
    const int blocks = c[0], threads = c[1], lds = c[2], iters = 30, nw = blocks * threads / 64;
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), lds, 0, din, d, drec, iters);
    CK(hipDeviceSynchronize());
    std::vector<Rec> r(nw);
    CK(hipMemcpy(r.data(), drec, nw * sizeof(Rec), hipMemcpyDeviceToHost));
    std::map<unsigned, int> per_cu, per_simd;
    long long tmin = r[0].t0, tend = r[0].t1;
    for (auto& q : r) { tmin = std::min(tmin, q.t0); tend = std::max(tend, q.t1); }
    std::vector<double> start, dur;
    for (auto& q : r) {
      const unsigned simd = (q.hw >> 4) & 3, cu = (q.hw >> 8) & 15, sh = (q.hw >> 12) & 1, se = (q.hw >> 13) & 7, xcc = q.xcc & 15;
      const unsigned cuid = (xcc << 12) | (se << 8) | (sh << 4) | cu;
      per_cu[cuid]++; per_simd[(cuid << 2) | simd]++;
      start.push_back((q.t0 - tmin) * 0.01); dur.push_back((q.t1 - q.t0) * 0.01);
    }
    std::sort(start.begin(), start.end()); std::sort(dur.begin(), dur.end());
    std::map<int, int> hc, hs;
    for (auto& p : per_cu) hc[p.second]++;
    for (auto& p : per_simd) hs[p.second]++;
    printf("grid %4d x %4d threads, %5d B dynamic LDS: total %.1f us; CUs used %zu, SIMDs used %zu\n", blocks, threads, lds, (tend - tmin) * 0.01,
           per_cu.size(), per_simd.size());
    printf("   waves per CU  :"); for (auto& p : hc) printf("  %d waves on %d CUs;", p.first, p.second); printf("\n");
    printf("   waves per SIMD:"); for (auto& p : hs) printf("  %d waves on %d SIMDs;", p.first, p.second); printf("\n");
    printf("   wave start (us after the first): median %.1f  p90 %.1f  max %.1f;  wave duration (us): min %.1f median %.1f max %.1f\n",
           start[nw / 2], start[nw * 9 / 10], start[nw - 1], dur[0], dur[nw / 2], dur[nw - 1]);
  }
  return 0;
}
