// Train-step shell and evaluation metrics for gfx950: the steps directly after
// the hot path in the reference loop.
//
//   core/tools/train.py:82-94   clip_grad_norm_(model.parameters(), cfg.train.clip_grad) ; optimizer.step()
//   core/tools/train.py:190-202 optim.SGD(lr, momentum, weight_decay) (+ MultiStepLR: host arithmetic only)
//   core/utils/metric.py:137-157 top-k correctness + confusion matrix of one batch of class scores
//
// Multi-tensor: every parameter tensor of the model rides in ONE launch (the table is a kernel argument, a
// workgroup finds its tensor with a scalar scan), 16 B per lane, HBM-bound: the update reads p, g, momentum and
// writes p, momentum = 20 B per element.  The gradient-norm is a two-level fixed-order reduction (per-workgroup
// partials, fp64 finalize), so the clipping coefficient never leaves the device and no float atomics are used.
#include "tbn_common.h"
#include "../../include/tbn_hip.h"

#define OPT_CHUNK 16384  // elements per workgroup

struct OptTab {
  int n;
  float* p[TBN_OPT_MAX_TENSORS];
  const float* g[TBN_OPT_MAX_TENSORS];
  float* m[TBN_OPT_MAX_TENSORS];
  unsigned long long len[TBN_OPT_MAX_TENSORS];
  int blk0[TBN_OPT_MAX_TENSORS + 1];
  unsigned char vec[TBN_OPT_MAX_TENSORS];   // all pointers of the tensor 16-byte aligned -> float4 path
};

static int fill_tab(OptTab* tab, const tbn_opt_tensor* t, int nt, bool need_p, bool need_m) {
  if (nt < 0 || nt > TBN_OPT_MAX_TENSORS) {
    tbn_set_error("optimizer: %d tensors per call (max %d)", nt, TBN_OPT_MAX_TENSORS);
    return TBN_ERR_ARG;
  }
  tab->n = nt;
  tab->blk0[0] = 0;
  for (int i = 0; i < nt; ++i) {
    if (t[i].grad == nullptr || (need_p && t[i].param == nullptr) || (need_m && t[i].momentum == nullptr)) {
      tbn_set_error("optimizer: tensor %d has a null pointer", i);
      return TBN_ERR_ARG;
    }
    if (((uintptr_t)t[i].grad | (uintptr_t)t[i].param | (uintptr_t)t[i].momentum) & 3u) {
      tbn_set_error("optimizer: tensor %d is not 4-byte aligned", i);
      return TBN_ERR_ARG;
    }
    // gradients that autograd hands out as slices of a larger tensor (e.g. of the concatenated class-head
    // weight) are only element aligned: those (small) tensors take the scalar path
    tab->vec[i] = ((((uintptr_t)t[i].grad | (uintptr_t)t[i].param | (uintptr_t)t[i].momentum) & 15u) == 0) ? 1 : 0;
    tab->p[i] = (float*)t[i].param;
    tab->g[i] = (const float*)t[i].grad;
    tab->m[i] = (float*)t[i].momentum;
    tab->len[i] = t[i].count;
    const size_t blocks = (t[i].count + OPT_CHUNK - 1) / OPT_CHUNK;
    if (blocks > (1u << 20) || tab->blk0[i] + (long)blocks > (1l << 30)) {
      tbn_set_error("optimizer: tensor %d too large", i);
      return TBN_ERR_ARG;
    }
    tab->blk0[i + 1] = tab->blk0[i] + (int)blocks;
  }
  return TBN_OK;
}

__device__ __forceinline__ int find_tensor(const OptTab& tab, int b) {
  int l = 0;
  while (l + 1 < tab.n && b >= tab.blk0[l + 1]) ++l;
  return l;
}

// ---------------------------------------------------------------- sum of squares, per-workgroup partials
__global__ __launch_bounds__(256) void opt_sqnorm_kernel(OptTab tab, float* __restrict__ partial) {
  __shared__ float red[4];
  const int l = find_tensor(tab, blockIdx.x);
  const size_t n = tab.len[l];
  const size_t beg = (size_t)(blockIdx.x - tab.blk0[l]) * OPT_CHUNK;
  const size_t end = beg + OPT_CHUNK < n ? beg + OPT_CHUNK : n;
  const float* g = tab.g[l];
  float s = 0.f;
  const size_t end4 = tab.vec[l] ? beg + ((end - beg) & ~(size_t)3) : beg;
  for (size_t i = beg + (size_t)threadIdx.x * 4; i < end4; i += 1024) {
    const float4 v = *reinterpret_cast<const float4*>(g + i);
    s = fmaf(v.x, v.x, s); s = fmaf(v.y, v.y, s); s = fmaf(v.z, v.z, s); s = fmaf(v.w, v.w, s);
  }
  for (size_t i = end4 + threadIdx.x; i < end; i += 256) s = fmaf(g[i], g[i], s);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// total_norm = sqrt(sum partials) (fp64, fixed order); coef = min(1, max_norm / (total_norm + 1e-6))
// (torch.nn.utils.clip_grad_norm_, norm_type 2)
__global__ __launch_bounds__(256) void opt_clip_coef_kernel(const float* __restrict__ partial, int nparts, float max_norm,
                                                            float* __restrict__ total_norm, float* __restrict__ coef) {
  __shared__ double red[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < nparts; i += 256) s += (double)partial[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float tn = (float)sqrt(red[0]);
    float c = max_norm / (tn + 1e-6f);
    if (c > 1.f) c = 1.f;
    total_norm[0] = tn;
    coef[0] = c;
  }
}

// ---------------------------------------------------------------- g *= coef (the reference clips in place)
__global__ __launch_bounds__(256) void opt_scale_kernel(OptTab tab, const float* __restrict__ coef) {
  const float c = coef[0];
  if (c == 1.f) return;  // torch multiplies by the clamped coefficient; x * 1 is exact, so skipping is identical
  const int l = find_tensor(tab, blockIdx.x);
  const size_t n = tab.len[l];
  const size_t beg = (size_t)(blockIdx.x - tab.blk0[l]) * OPT_CHUNK;
  const size_t end = beg + OPT_CHUNK < n ? beg + OPT_CHUNK : n;
  float* g = const_cast<float*>(tab.g[l]);
  const size_t end4 = tab.vec[l] ? beg + ((end - beg) & ~(size_t)3) : beg;
  for (size_t i = beg + (size_t)threadIdx.x * 4; i < end4; i += 1024) {
    float4 v = *reinterpret_cast<float4*>(g + i);
    v.x *= c; v.y *= c; v.z *= c; v.w *= c;
    *reinterpret_cast<float4*>(g + i) = v;
  }
  for (size_t i = end4 + threadIdx.x; i < end; i += 256) g[i] *= c;
}

// ---------------------------------------------------------------- SGD with momentum (torch.optim.SGD, dampening 0)
//   d = g * gscale + wd * p ; m = momentum * m + d ; p -= lr * m          (m starts at 0: first step m = d)
__global__ __launch_bounds__(256) void opt_sgd_kernel(OptTab tab, float lr, float momentum, float wd,
                                                      const float* __restrict__ gscale) {
  const float gs = gscale != nullptr ? gscale[0] : 1.f;
  const int l = find_tensor(tab, blockIdx.x);
  const size_t n = tab.len[l];
  const size_t beg = (size_t)(blockIdx.x - tab.blk0[l]) * OPT_CHUNK;
  const size_t end = beg + OPT_CHUNK < n ? beg + OPT_CHUNK : n;
  float* p = tab.p[l];
  const float* g = tab.g[l];
  float* m = tab.m[l];
  auto upd = [&](float pv, float gv, float mv, float& pn, float& mn) {
    float d = gs == 1.f ? gv : gv * gs;
    if (wd != 0.f) d = fmaf(wd, pv, d);   // torch: d_p.add(p, alpha=wd)
    mn = momentum != 0.f ? fmaf(momentum, mv, d) : d;
    pn = fmaf(-lr, mn, pv);               // torch: p.add_(buf, alpha=-lr)
  };
  const size_t end4 = tab.vec[l] ? beg + ((end - beg) & ~(size_t)3) : beg;
  for (size_t i = beg + (size_t)threadIdx.x * 4; i < end4; i += 1024) {
    const float4 pv = tbn_ld4<(TBN_BN_NT & 8) != 0>(p + i);
    const float4 gv = tbn_ld4<(TBN_BN_NT & 8) != 0>(g + i);
    float4 mv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (m != nullptr) mv = tbn_ld4<(TBN_BN_NT & 8) != 0>(m + i);
    float4 pn, mn;
    upd(pv.x, gv.x, mv.x, pn.x, mn.x);
    upd(pv.y, gv.y, mv.y, pn.y, mn.y);
    upd(pv.z, gv.z, mv.z, pn.z, mn.z);
    upd(pv.w, gv.w, mv.w, pn.w, mn.w);
    *reinterpret_cast<float4*>(p + i) = pn;
    if (m != nullptr) *reinterpret_cast<float4*>(m + i) = mn;
  }
  for (size_t i = end4 + threadIdx.x; i < end; i += 256) {
    float pn, mn;
    upd(p[i], g[i], m != nullptr ? m[i] : 0.f, pn, mn);
    p[i] = pn;
    if (m != nullptr) m[i] = mn;
  }
}

extern "C" {

int tbn_opt_num_partials(const tbn_opt_tensor* tensors, int num_tensors) {
  long b = 0;
  if (tensors == nullptr || num_tensors < 0) return 0;
  for (int i = 0; i < num_tensors; ++i) b += (long)((tensors[i].count + OPT_CHUNK - 1) / OPT_CHUNK);
  return b > (1l << 30) ? 0 : (int)b;
}

int tbn_opt_sqnorm_partials(const tbn_opt_tensor* tensors, int num_tensors, float* partials, void* stream) {
  TBN_REQUIRE(tensors != nullptr && partials != nullptr, "opt_sqnorm_partials: null argument");
  OptTab tab;
  int rc = fill_tab(&tab, tensors, num_tensors, false, false);
  if (rc != TBN_OK) return rc;
  if (tab.blk0[tab.n] == 0) return TBN_OK;
  TBN_KLAUNCH(opt_sqnorm_kernel, dim3(tab.blk0[tab.n]), dim3(256), 0, (hipStream_t)stream, tab, partials);
  TBN_CHECK_LAUNCH("opt_sqnorm");
  return TBN_OK;
}

int tbn_opt_clip_coef(const float* partials, int num_partials, float max_norm, float* total_norm, float* coef,
                      void* stream) {
  TBN_REQUIRE(partials != nullptr && total_norm != nullptr && coef != nullptr && num_partials >= 0,
              "opt_clip_coef: bad argument");
  TBN_KLAUNCH(opt_clip_coef_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partials, num_partials, max_norm,
                     total_norm, coef);
  TBN_CHECK_LAUNCH("opt_clip_coef");
  return TBN_OK;
}

int tbn_opt_scale_grads(const tbn_opt_tensor* tensors, int num_tensors, const float* coef, void* stream) {
  TBN_REQUIRE(tensors != nullptr && coef != nullptr, "opt_scale_grads: null argument");
  OptTab tab;
  int rc = fill_tab(&tab, tensors, num_tensors, false, false);
  if (rc != TBN_OK) return rc;
  if (tab.blk0[tab.n] == 0) return TBN_OK;
  TBN_KLAUNCH(opt_scale_kernel, dim3(tab.blk0[tab.n]), dim3(256), 0, (hipStream_t)stream, tab, coef);
  TBN_CHECK_LAUNCH("opt_scale_grads");
  return TBN_OK;
}

int tbn_opt_sgd_step(const tbn_opt_tensor* tensors, int num_tensors, float lr, float momentum, float weight_decay,
                     const float* grad_scale, void* stream) {
  TBN_REQUIRE(tensors != nullptr, "opt_sgd_step: null argument");
  OptTab tab;
  int rc = fill_tab(&tab, tensors, num_tensors, true, momentum != 0.f);
  if (rc != TBN_OK) return rc;
  if (tab.blk0[tab.n] == 0) return TBN_OK;
  TBN_KLAUNCH(opt_sgd_kernel, dim3(tab.blk0[tab.n]), dim3(256), 0, (hipStream_t)stream, tab, lr, momentum,
                     weight_decay, grad_scale);
  TBN_CHECK_LAUNCH("opt_sgd_step");
  return TBN_OK;
}

}  // extern "C"

// ---------------------------------------------------------------- metrics: top-k correctness + confusion matrix
// one wave per sample: k rounds of (max value, lowest index) wavefront arg-max over the class scores (the order
// torch.topk(sorted=True) returns for distinct scores; ties resolve to the lower class index).
// correct[j][b] = (j-th prediction of sample b == target[b]);  conf[target][top-1] += 1 (integer-valued float
// atomics: exact and order independent).
__global__ __launch_bounds__(256) void topk_correct_kernel(const float* __restrict__ scores, int ld,
                                                           const long long* __restrict__ target, int B, int C, int K,
                                                           unsigned char* __restrict__ correct,
                                                           long long* __restrict__ pred, float* __restrict__ conf) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const float* row = scores + (size_t)b * ld;
  const long long tgt = target[b];
  float prev_v = INFINITY;
  int prev_i = -1;
  for (int j = 0; j < K; ++j) {
    // best element strictly after (prev_v, prev_i) in (value desc, index asc) order
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int c = lane; c < C; c += 64) {
      const float v = row[c];
      const bool after = (v < prev_v) || (v == prev_v && c > prev_i);
      if (after && (v > bv || (v == bv && c < bi))) {
        bv = v;
        bi = c;
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(bv, o);
      const int oi = __shfl_xor(bi, o);
      if (ov > bv || (ov == bv && oi < bi)) {
        bv = ov;
        bi = oi;
      }
    }
    if (lane == 0) {
      correct[(size_t)j * B + b] = (unsigned char)((long long)bi == tgt);
      if (pred != nullptr) pred[(size_t)j * B + b] = bi;
      if (j == 0 && conf != nullptr && tgt >= 0 && tgt < C && bi < C) atomicAdd(conf + (size_t)tgt * C + bi, 1.0f);
    }
    prev_v = bv;
    prev_i = bi;
  }
}

extern "C" int tbn_topk_correct(const float* scores, int scores_ld, const long long* target, int batch, int classes,
                                int k, unsigned char* correct, long long* pred, float* conf_mat, void* stream) {
  TBN_REQUIRE(scores != nullptr && target != nullptr && correct != nullptr, "topk_correct: null argument");
  TBN_REQUIRE(batch >= 0 && classes >= 1 && k >= 1 && k <= classes && scores_ld >= classes,
              "topk_correct: bad shape (B=%d, C=%d, k=%d)", batch, classes, k);
  if (batch == 0) return TBN_OK;
  TBN_KLAUNCH(topk_correct_kernel, dim3(cdiv(batch, 4)), dim3(256), 0, (hipStream_t)stream, scores, scores_ld,
                     target, batch, classes, k, correct, pred, conf_mat);
  TBN_CHECK_LAUNCH("topk_correct");
  return TBN_OK;
}
