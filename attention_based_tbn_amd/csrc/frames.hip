// On-device input pipeline for the visual modalities: the step right before the hot path.
//
// Reference (host, per sample, NumPy + cv2):  core/utils/create_dataloader.py:19-81 composes
//   train:  MultiScaleCrop (crop box -> cv2.resize INTER_LINEAR to the input size)  ->  RandomHorizontalFlip
//   test :  Rescale (cv2.resize, smaller edge to test_scale_size)  ->  CenterCrop
//   both :  Stack (Flow: `length` single-channel images become the channels of one sample)  ->  ToTensor
//           (uint8 HWC -> float CHW, / 255)  ->  Normalize ((x - mean) / std)       core/dataset/transform.py:9-543
// Here the random decisions stay on the host (same NumPy draws), the pixels are touched ONCE: a single kernel reads
// the uint8 frames, resizes / crops / flips on the fly and writes the normalised fp32 NCHW tensor the model takes.
//
// Resize arithmetic = OpenCV's 8-bit INTER_LINEAR (imgproc/resize.cpp: 11-bit fixed-point weights, cvRound,
//   horizontal pass in int, vertical pass ((b0*(S0>>4))>>16 + (b1*(S1>>4))>>16 + 2) >> 2).  cv2 is NOT in the build
//   image, so that part is restated from the published algorithm ("parity unpinned"); crop / flip / stack /
//   ToTensor / Normalize are pinned bit-exactly to the reference classes.
#include "tbn_common.h"
#include "../../include/tbn_hip.h"

struct FramesP {
  const unsigned char* src;
  float* out;
  int n_img, H, W, C;            // source frames (n_img, H, W, C) uint8
  int bx, by, bw, bh;            // box of the source that is resized
  int rw, rh;                    // size the box is resized to (== bw, bh: no interpolation)
  int cx, cy, ow, oh;            // crop window inside the resized box = output size
  int flip, stack, div255;
  int n_stat;                    // entries of mean / std (repeated over the channels, as Normalize does)
  double scale_x, scale_y;
};

__device__ __forceinline__ int cv_round_to_short(float v) {
  // saturate_cast<short>(cvRound(v)): round half to even; v is in [0, 2048]
  return (int)__float2int_rn(v);
}

__device__ __forceinline__ void lin_coef(int d, double scale, int ssize, int* s0, int* a0, int* a1, bool clamp_frac) {
  float f = (float)((d + 0.5) * scale - 0.5);
  int s = (int)floorf(f);
  f -= (float)s;
  if (clamp_frac) {                    // x direction: the fraction is zeroed at the borders
    if (s < 0) { f = 0.f; s = 0; }
    if (s >= ssize - 1) { f = 0.f; s = ssize - 1; }
  }
  *s0 = s;
  *a0 = cv_round_to_short((1.f - f) * 2048.f);
  *a1 = cv_round_to_short(f * 2048.f);
}

// One workgroup = kRows output rows of ONE output plane (sample n, channel c).  A thread owns the columns x = tid,
// tid + 256, ...: their horizontal source positions / weights are computed once and reused for every row; the rows'
// vertical coefficients are computed once per workgroup (LDS).  Per output element that leaves four byte loads, the
// fixed-point blend, one table look-up (ToTensor / Normalize) and one coalesced store -- the first version decoded
// the flat index (64-bit div / mod chain) and both coefficient pairs (double precision) per ELEMENT and ran at
// 1.0-1.35 TB/s of algorithmic bytes.
constexpr int kFrameRows = 32;
constexpr int kFrameCols = 4;     // columns per thread: output widths up to 1024
__global__ __launch_bounds__(256) void frames_to_tensor_kernel(FramesP p, const float* __restrict__ mean,
                                                               const float* __restrict__ stdv) {
  __shared__ int ysrc0[kFrameRows], ysrc1[kFrameRows], yw0[kFrameRows], yw1[kFrameRows];
  const int co = p.C * p.stack;                       // output channels per sample
  const int row_tiles = (p.oh + kFrameRows - 1) / kFrameRows;
  const int plane = blockIdx.x / row_tiles, rt = blockIdx.x - plane * row_tiles;
  const int n = plane / co, c = plane - n * co;
  const int img = n * p.stack + c / p.C, ch = c % p.C;
  const int y_begin = rt * kFrameRows, y_end = min(p.oh, y_begin + kFrameRows);
  const bool resize = (p.rw != p.bw) || (p.rh != p.bh);
  const int tid = threadIdx.x;
  if (tid < y_end - y_begin) {
    const int yr = p.cy + y_begin + tid;              // position in the resized box
    if (resize) {
      int sy, ay0, ay1;
      lin_coef(yr, p.scale_y, p.bh, &sy, &ay0, &ay1, false);
      ysrc0[tid] = p.by + min(max(sy, 0), p.bh - 1);     // rows are clamped
      ysrc1[tid] = p.by + min(max(sy + 1, 0), p.bh - 1);
      yw0[tid] = ay0;
      yw1[tid] = ay1;
    } else {
      ysrc0[tid] = ysrc1[tid] = p.by + yr;
      yw0[tid] = yw1[tid] = 0;
    }
  }
  int xs0[kFrameCols], xs1[kFrameCols], xa0[kFrameCols], xa1[kFrameCols];
#pragma unroll
  for (int k = 0; k < kFrameCols; ++k) {
    const int x = tid + 256 * k;
    xs0[k] = xs1[k] = xa0[k] = xa1[k] = 0;
    if (x < p.ow) {
      const int xr = p.cx + (p.flip ? p.ow - 1 - x : x);
      if (resize) {
        int sx, ax0, ax1;
        lin_coef(xr, p.scale_x, p.bw, &sx, &ax0, &ax1, true);
        xs0[k] = (p.bx + sx) * p.C;
        xs1[k] = (p.bx + min(sx + 1, p.bw - 1)) * p.C;   // weight 0 when clamped
        xa0[k] = ax0;
        xa1[k] = ax1;
      } else {
        xs0[k] = (p.bx + xr) * p.C;
      }
    }
  }
  // ToTensor / Normalize see only 256 different inputs per plane: their exact (correctly rounded, as torch) divisions
  // are done once per workgroup, the pixel loop looks the result up
  __shared__ float lut[256];
  {
    float f = (float)tid;
    if (p.div255) f = f / 255.f;
    if (p.n_stat > 0) f = (f - mean[c % p.n_stat]) / stdv[c % p.n_stat];
    lut[tid] = f;
  }
  const unsigned char* __restrict__ base = p.src + (size_t)img * p.H * p.W * p.C + ch;
  float* __restrict__ out = p.out + ((size_t)plane * p.oh + y_begin) * p.ow;
  __syncthreads();
  for (int y = 0; y < y_end - y_begin; ++y) {
    const unsigned char* r0 = base + (size_t)ysrc0[y] * p.W * p.C;
    const unsigned char* r1 = base + (size_t)ysrc1[y] * p.W * p.C;
    const int ay0 = yw0[y], ay1 = yw1[y];
#pragma unroll
    for (int k = 0; k < kFrameCols; ++k) {
      const int x = tid + 256 * k;
      if (x >= p.ow) break;
      int v;
      if (!resize) {
        v = r0[xs0[k]];
      } else {
        const int h0 = (int)r0[xs0[k]] * xa0[k] + (int)r0[xs1[k]] * xa1[k];
        const int h1 = (int)r1[xs0[k]] * xa0[k] + (int)r1[xs1[k]] * xa1[k];
        v = (((ay0 * (h0 >> 4)) >> 16) + ((ay1 * (h1 >> 4)) >> 16) + 2) >> 2;
        v = min(max(v, 0), 255);
      }
      out[(size_t)y * p.ow + x] = lut[v];
    }
  }
}

extern "C" int tbn_frames_to_tensor(const unsigned char* frames, int n_img, int height, int width, int channels,
                                    int box_x, int box_y, int box_w, int box_h, int resized_w, int resized_h,
                                    int crop_x, int crop_y, int out_w, int out_h, int flip, int stack,
                                    const float* mean, const float* std_dev, int n_stat, int div255, float* out,
                                    void* stream) {
  TBN_REQUIRE(frames != nullptr && out != nullptr, "frames_to_tensor: null argument");
  TBN_REQUIRE(n_img >= 0 && height > 0 && width > 0 && channels >= 1 && channels <= 4 && stack >= 1 &&
                  n_img % stack == 0,
              "frames_to_tensor: bad frame stack (n=%d, %dx%dx%d, stack %d)", n_img, height, width, channels, stack);
  TBN_REQUIRE(box_x >= 0 && box_y >= 0 && box_w > 0 && box_h > 0 && box_x + box_w <= width && box_y + box_h <= height,
              "frames_to_tensor: source box (%d,%d,%d,%d) outside the %dx%d frame", box_x, box_y, box_w, box_h, width,
              height);
  TBN_REQUIRE(resized_w > 0 && resized_h > 0 && crop_x >= 0 && crop_y >= 0 && out_w > 0 && out_h > 0 &&
                  crop_x + out_w <= resized_w && crop_y + out_h <= resized_h,
              "frames_to_tensor: crop window (%d,%d,%d,%d) outside the resized %dx%d box", crop_x, crop_y, out_w, out_h,
              resized_w, resized_h);
  TBN_REQUIRE(n_stat == 0 || (mean != nullptr && std_dev != nullptr), "frames_to_tensor: mean/std missing");
  if (n_img == 0) return TBN_OK;
  FramesP p;
  p.src = frames; p.out = out;
  p.n_img = n_img; p.H = height; p.W = width; p.C = channels;
  p.bx = box_x; p.by = box_y; p.bw = box_w; p.bh = box_h;
  p.rw = resized_w; p.rh = resized_h;
  p.cx = crop_x; p.cy = crop_y; p.ow = out_w; p.oh = out_h;
  p.flip = flip ? 1 : 0; p.stack = stack; p.div255 = div255 ? 1 : 0; p.n_stat = n_stat;
  // resize.cpp: inv_scale = dsize / ssize (double); scale = 1. / inv_scale
  p.scale_x = 1.0 / ((double)resized_w / (double)box_w);
  p.scale_y = 1.0 / ((double)resized_h / (double)box_h);
  TBN_REQUIRE(out_w <= 256 * kFrameCols, "frames_to_tensor: output width %d > %d", out_w, 256 * kFrameCols);
  const size_t planes = (size_t)(n_img / stack) * channels * stack;
  const size_t g = planes * (size_t)((out_h + kFrameRows - 1) / kFrameRows);
  TBN_REQUIRE(g < (1ull << 31), "frames_to_tensor: too many output planes");
  TBN_KLAUNCH(frames_to_tensor_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, p, mean, std_dev);
  TBN_CHECK_LAUNCH("frames_to_tensor");
  return TBN_OK;
}
