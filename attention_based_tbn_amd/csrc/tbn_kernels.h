// Internal launch-level interface between the HIP kernels and the C-ABI / backbone engine.
#pragma once
#include "tbn_common.h"
#include <hip/hip_ext.h>

enum { CONV_EPI_PLAIN = 0, CONV_EPI_STATS = 1, CONV_EPI_EVAL = 2 };
enum {
  CONV_FLAG_ACCUM = 1,
  CONV_FLAG_RELU = 2,
  CONV_FLAG_HALO = 4,   // host only: use the LDS-halo kernel
  CONV_FLAG_DMA = 8,    // host only: use the LDS-DMA staging kernel
  CONV_FLAG_SK4 = 16    // host only: 32-row tiles whose four waves split K (small-M launches)
};

// BN-backward reduce fused into a data-gradient epilogue (the launch that writes the FINAL value of dz): for the output
// columns [col_begin, col_begin + C) -- one BN layer of the producer side -- the epilogue forms
//   g = dz * [y*scale+shift > 0],  S1 = sum g,  S2 = sum g * (y-mean)*rstd
// per M tile and writes them as partial[(tile*2 + {0,1}) * C + (col - col_begin)] (what bn_bwd_reduce would have
// produced from a second pass over dz and y).  y == nullptr: columns without a BN (pass-through pool slice).
#define TBN_CONV_MAXSEG 4
struct RedSeg {
  const float* y;         // BN input of that layer at column col_begin (pitch y_ld), same pixel order as the output
  float* partial;
  const float* stats;     // mean at stats[c], rstd at stats[chan + c], scale at stats[2*chan + c], shift at stats[3*chan + c]
  int y_ld, col_begin, C, c_off;
  unsigned y_bytes;
};

struct ConvP {
  const float* in;
  const float* wt;
  const float* bias;      // may be null
  const float* scale;     // EVAL epilogue
  const float* shift;     // EVAL epilogue
  float* stat_partial;    // STATS epilogue: [tiles_m][2][Cout]
  Seg seg[TBN_CONV_MAXSEG];
  int nseg;
  int raw_seg1;           // EVAL epilogue: 1 + index of a segment written as the bare accumulator (no BN fold / ReLU); 0 = none
  int in_ld;
  // fused BN-backward reduce (PLAIN epilogue of a data gradient), see RedSeg
  RedSeg red[TBN_CONV_MAXSEG];
  int nred;               // 0 = off
  int red_chan;           // stride between the mean / rstd / scale / shift arrays
  int red_row0;           // first partial row of this launch (parity phases of one layer share a partial buffer)
  // ---- caller-facing geometry (host side fills the derived fields below from these)
  int N, H, W;            // input spatial dims (for dgrad: dims of dy)
  int OH, OW;             // full output dims
  int Cin;                // K per tap (ROWMODE: floats of one contiguous input run; R runs make the K of a pixel)
  int Cout;
  int R, S, stride, pad, up;
  int M, K;               // GEMM M (output pixels of this launch) and reduction length of this launch
  int cp;                 // ROWMODE: floats per pixel of the (physically zero-padded) input image
  int tiles_m, tiles_n;
  int mode, flags;
  double alg_flops;       // host only: algorithmic FLOPs of this launch (profiling)
  int stages;             // host only: LDS stages (1 = two barriers per K-step, 2 = double buffered); 0 = default
  int halo;               // host only: 1 = LDS-halo kernel (3x3 / stride 1 / pad 1 layers: input patch staged once per chunk),
                          //            2 = LDS-DMA staging of the generic kernel, 3 = 32-row tiles with the waves splitting K
                          //            (statistics / reduce partial rows are then per 32*mt rows, not 128*mt)
  // ---- derived by tbn_launch_conv
  unsigned in_bytes, wt_bytes;          // buffer extents: out-of-range lanes read zeros (hardware check)
  unsigned seg_bytes[TBN_CONV_MAXSEG];  // extent of each output segment from seg[i].ptr
  int OHs, OWs;                         // output sub-grid this launch covers (m -> n, a, b)
  int out_sy, out_oy, out_sx, out_ox;   // full-grid output pixel = (a*out_sy+out_oy, b*out_sx+out_ox)
  int in_sy, in_sx;                     // input step per sub-grid step
  int Krow;                             // floats per weight row (all taps)
  int ntaps;
  int tap_dy[9], tap_dx[9];             // input offset of tap t (padding folded in)
  int tap_koff[9];                      // float offset of tap t inside a weight row
  int tap_off[9];                       // byte offset of tap t relative to the window origin pixel
  int ty0, tny, tx0, tnx;               // the taps form a grid: tap (ry, rx) = (ty0 + ry, tx0 + rx), t = ry*tnx + rx
  FastDiv div_ohw, div_ow;              // m -> (n, a, b) without integer division
  FastDiv div_rl4;                      // ROWMODE: float4 index inside K -> (run, float4 inside the run)
  // -DTBN_DIAG=1 builds only (scripts/README.md, TBN_DIAG_FOLD): per-input-channel scale / shift applied with a ReLU when
  // the LDS-halo kernel stages its patch -- the COST side of folding the producer's BN apply into this conv's operand
  // staging (the engine passes ones / zeros: results unchanged); null = off, never read by the shipped build
  const float* fold_scale;
  const float* fold_shift;
};

// ---- riders: elementwise BN passes of INDEPENDENT layers executed by extra workgroups of a conv-GEMM launch.
// Inside an inception block (reference dataflow core/models/bn_inception_audio.py:437-1003, concat :485-493) the BN apply
// of the `1x1` column range does not feed the `3x3 | double_3x3_1` GEMMs, the BN apply of `3x3` / `pool_proj` does not
// feed `double_3x3_2`, and in backward the BN-backward apply of `1x1` / `3x3` / `pool_proj` does not feed the data gradient
// of `double_3x3_2`.  Cross-stream dependencies are the expensive primitive on this runtime (DESIGN.md finding 15), so the
// concurrency comes from ONE launch: the grid of the GEMM is extended by `span` workgroups that run the HBM-bound pass
// (same device code as the stand-alone bn_*_multi kernels: bit-identical results) beside the MFMA-bound tiles.
#define TBN_RIDER_MAXL 3
struct RiderLayer {
  const float* y;       // BN input rows (pitch y_ld)
  const float* dz;      // kind 2: gradient wrt z (pitch dz_ld)
  float* out;           // kind 1: z (pitch out_ld); kind 2: dy (pitch y_ld, in place over y)
  const float* scale;
  const float* shift;
  const float* coef;    // kind 2: [3][C]
  int y_ld, dz_ld, out_ld, P, C, rows;   // rows = pixel rows per rider workgroup
};
struct RiderP {
  int span;     // grid slots reserved for the rider (0: none)
  int first;    // first grid slot of the rider: 0 (front of the grid) or the GEMM's workgroup count (behind it)
  int gemm0;    // grid slot of GEMM workgroup 0 (front placement: span, a multiple of 8 -- keeps the XCD-aware tile maps)
  int nblk;     // rider workgroups that do work (<= span)
  int kind;     // 1: z = relu(y * scale + shift);  2: dy = a * [z > 0] dz + b * y + c
  int n;
  int blk0[TBN_RIDER_MAXL + 1];
  RiderLayer l[TBN_RIDER_MAXL];
};
// places a rider with `nblk` workgroups around a GEMM grid of `gemm_blocks`; returns the launch's grid size
int tbn_rider_place(RiderP* r, int gemm_blocks);

struct WgradP {
  const float* dy;
  const float* x;
  float* out;
  int dy_ld, x_ld;
  int N, H, W, OH, OW;
  int Cin, Cout;          // per tap (ROWMODE: taps = 1 and Cin = the whole packed K of a pixel = R runs of `rl` floats)
  int R, S, stride, pad;
  int taps;
  int M, K;
  int cp, rl;             // ROWMODE: floats per pixel of the padded input image / floats per contiguous run
  int tiles_co, tiles_ci;
  int rows_per_split;
  int mt, nt;             // host only: tile (32-column sub-tiles of Cout / Cin per workgroup), 0 = heuristic
  FastDiv div_ohw, div_ow, div_rl4;
  // incremental row addressing of the kernel (filled by tbn_launch_wgrad): a lane's row advances 64 output pixels per step
  unsigned mul_ow;        // ceil(2^32 / OW): oy = umulhi(pp, mul_ow) for the in-frame pixel index pp (pp * OW < 2^32); 0 when OW == 1
  unsigned r64;           // 64 % (OH*OW)
  unsigned frame_bytes;   // bytes of one input frame (H * W * pitch)
  unsigned fb_lo, fb_hi;  // frame-base advance per step: (64 / (OH*OW)) frames, one more when pp wraps
  unsigned row_step, col_step;   // ROWMODE: bytes per output row / column step in the padded image (stride folded in)
  double alg_flops;       // host only
  unsigned dy_bytes, x_bytes;   // exact extents (last row ends at its last column): lanes beyond them read zeros
  const float* fold_scale;      // -DTBN_DIAG=1 builds only: as ConvP::fold_scale, applied to the staged x operand
  const float* fold_shift;
  int ablate;             // timing ablations, honoured by -DTBN_ABLATE=1 builds only (scripts/wgrad_ablate.py)
};

// optional in-process profiler: every conv-GEMM launch gets a pair of hipEvents on its stream.  The events ride ON the
// kernel's dispatch packet (hipExtLaunchKernelGGL: begin / end timestamps of the kernel itself, what rocprofv3's kernel
// trace reports); rounds 1-2 recorded two separate marker packets around the launch, whose processing sat inside the
// bracket (HIP-event averages 2.8 % above rocprofv3's for the same launches, round-2 verdict).
void tbn_prof_begin(const char* kernel, double flops, hipStream_t st, double alg_bytes = 0.0);
void tbn_prof_end(hipStream_t st);
void tbn_prof_label(const char* label);
bool tbn_prof_launch_events(hipEvent_t* start, hipEvent_t* stop);   // true: an open profiler record wants this launch timed
#define TBN_LAUNCH(kernel, grid, block, lds, st, ...)                                            \
  do {                                                                                           \
    hipEvent_t tbn_e0__, tbn_e1__;                                                               \
    if (tbn_prof_launch_events(&tbn_e0__, &tbn_e1__))                                            \
      hipExtLaunchKernelGGL(kernel, grid, block, lds, st, tbn_e0__, tbn_e1__, 0, __VA_ARGS__);  \
    else                                                                                         \
      TBN_KLAUNCH(kernel, grid, block, lds, st, __VA_ARGS__);                                    \
  } while (0)

// conv_igemm.hip
void tbn_conv_pick_tile(int M, int Cout, int K, int* mt, int* nt);
int tbn_launch_conv(ConvP p, int rowmode, int mt, int nt, hipStream_t st, const RiderP* rider = nullptr);
// two independent unit-stride convs in one launch (variant 0 LDS-halo, 1 / 2 generic with 1 / 2 LDS stages; tiles <= (2,2))
int tbn_launch_conv_pair(ConvP a, ConvP b, int variant, int mt, int nt, hipStream_t st, const RiderP* rider = nullptr);
int tbn_conv_red_rows(int N, int OH, int OW, int up, int tile_rows);   // tile_rows = M rows per workgroup tile (128 * mt; 32 * mt for the split-K tile kernel)
size_t tbn_conv_halo_lds_bytes(const ConvP& p, int mt, int nt);   // 0: shape not handled by the LDS-halo kernel
void tbn_wgrad_plan(int M, int Cout, int Cin, int taps, int* mt, int* nt, int* splits, int* rows_per_split);
size_t tbn_wgrad_workspace_floats(int M, int Cout, int Cin, int taps);
int tbn_launch_wgrad(WgradP p, int rowmode, float* dw, float* workspace, hipStream_t st);
int tbn_launch_weight_flip_transpose(const float* w, float* wt, int Cout, int taps, int Cin, hipStream_t st);
// layer table of the one-launch flip/transpose of every data-gradient weight of a backbone (kernel argument)
struct FlipTab {
  int n;
  int w_off[64];     // float offset of the layer inside the flat weight array (same offset in the flipped copy)
  short cout[64], cin[64], taps[64];
  int blk0[65];      // first workgroup of each layer; blk0[n] = grid size
};
int tbn_launch_weight_flip_transpose_all(const float* w, float* wt, const FlipTab& tab, hipStream_t st);

// bn.hip
int tbn_launch_bn_stats(const float* y, int ld, int P, int C, float* partial, int* nparts, hipStream_t st);
int tbn_bn_stats_parts(int P, int C);
int tbn_launch_bn_finalize(const float* partial, int nparts, int P, int C, const float* gamma, const float* beta,
                           const float* conv_bias, float* running_mean, float* running_var, float momentum, float eps,
                           float* save_mean, float* save_rstd, float* scale, float* shift, hipStream_t st);
int tbn_launch_bn_apply(const float* y, int P, int C, const float* scale, const float* shift, const Seg* segs,
                        int nseg, hipStream_t st);
int tbn_launch_bn_apply_maxpool(const float* y, int N, int H, int W, int C, const float* scale, const float* shift,
                                float* out, int out_ld, unsigned char* argmax, int OH, int OW, int stride, int pad,
                                hipStream_t st);
int tbn_launch_bn_bwd_reduce_pooled(const float* dpooled, int dpooled_ld, const unsigned char* argmax, int N, int H,
                                    int W, int OH, int OW, int stride, int pad, const float* y, int C,
                                    const float* scale, const float* shift, const float* mean, const float* rstd,
                                    float* partial, hipStream_t st);
int tbn_bn_bwd_pooled_parts(int N, int H, int W, int C, int stride, int pad);   // partial rows of the pooled reduce
int tbn_launch_bn_bwd_apply_pooled(const float* dpooled, int dpooled_ld, const unsigned char* argmax, int N, int H, int W,
                                   int OH, int OW, int stride, int pad, const float* y, int C, const float* scale,
                                   const float* shift, const float* coef, float* dy, hipStream_t st);
// batched training-BN launches over up to three independent layers (bn_multi.hip)
#define TBN_BN_MAXL 4
struct BnFwdLayer {
  const float* y;           // BN input (P, C) with pitch y_ld (a column range of a wider conv output)
  int y_ld;
  int P, C;
  const float* partial;     // statistics partials: partial[(i*2 + {0,1}) * pld + c], i < nparts
  int pld, nparts;
  const float *gamma, *beta, *conv_bias;
  float *running_mean, *running_var, *save_mean, *save_rstd, *scale, *shift;
  Seg seg[3];               // destination column ranges of z
  int nseg;
  int app_rows;             // filled by the launcher: rows per workgroup of the apply kernel
};
struct BnFwdBatch {
  int n;
  float momentum, eps;
  BnFwdLayer l[TBN_BN_MAXL];
  int fin_blk0[TBN_BN_MAXL + 1], app_blk0[TBN_BN_MAXL + 1];   // filled by the launcher
};
struct BnBwdLayer {
  CSeg dz[3];               // gradient wrt z, by column range
  int nseg;
  const float* y;           // BN input (pitch y_ld); dy is written in place
  float* dy;
  int y_ld;
  int P, C;
  const float *scale, *shift, *mean, *rstd;
  float* partial;           // [nparts][2][C]: scratch of the reduce kernel, or -- ext_parts > 0 -- already filled by the
  int ext_parts;            //   data-gradient epilogue that finished dz (conv RedSeg): no reduce pass for this layer
  float* coef;              // scratch [3][C]
  float *dgamma, *dbeta, *dbias;
  int pch, nparts;          // filled by the launcher
  int app_rows;             // filled by the launcher: rows per workgroup of the apply kernel
};
struct BnBwdBatch {
  int n;
  BnBwdLayer l[TBN_BN_MAXL];
  int red_blk0[TBN_BN_MAXL + 1], fin_blk0[TBN_BN_MAXL + 1], app_blk0[TBN_BN_MAXL + 1];   // filled by the launcher
};
int tbn_launch_bn_fwd_multi(BnFwdBatch& b, hipStream_t st);   // finalize + apply of every layer: 2 launches
int tbn_launch_bn_bwd_multi(BnBwdBatch& b, hipStream_t st);   // reduce + finalize + apply: 3 launches
// the same with some members' APPLY pass taken out (bit i of `defer_mask`): those are described in `*rider` instead (to be
// executed inside a later conv launch: tbn_launch_conv(..., rider)); every member is still reduced / finalized here
int tbn_launch_bn_fwd_multi_defer(BnFwdBatch& b, unsigned defer_mask, RiderP* rider, hipStream_t st);
int tbn_launch_bn_bwd_multi_defer(BnBwdBatch& b, unsigned defer_mask, RiderP* rider, hipStream_t st);
bool tbn_prof_enabled();   // the opt-in profiler brackets every conv launch: riders would be charged to their hosts
int tbn_launch_bn_fold(const float* gamma, const float* beta, const float* mean, const float* var, const float* bias,
                       float eps, float* scale, float* shift, int C, hipStream_t st);
int tbn_bn_bwd_parts(int P, int C);
int tbn_launch_bn_bwd_reduce(const CSeg* dz, int nseg, const float* y, int P, int C, const float* scale,
                             const float* shift, const float* mean, const float* rstd, float* partial, hipStream_t st);
int tbn_launch_bn_bwd_finalize(const float* partial, int nparts, int P, int C, const float* scale, const float* mean,
                               const float* rstd, float* coef, float* dgamma, float* dbeta, float* dbias,
                               hipStream_t st);
int tbn_launch_bn_bwd_apply(const CSeg* dz, int nseg, const float* y, int P, int C, const float* scale,
                            const float* shift, const float* coef, float* dy, hipStream_t st);

// pool.hip
int tbn_launch_maxpool_fwd(const float* in, int in_ld, float* out, int out_ld, uint8_t* argmax, int N, int H, int W,
                           int C, int OH, int OW, int stride, int pad, hipStream_t st);
int tbn_launch_maxpool_bwd(const float* dout, int dout_ld, const uint8_t* argmax, float* din, int din_ld, int N,
                           int H, int W, int C, int OH, int OW, int stride, int pad, int accumulate, hipStream_t st);
int tbn_launch_avgpool3_fwd(const float* in, int in_ld, float* out, int out_ld, int N, int H, int W, int C,
                            int accumulate, hipStream_t st);
int tbn_launch_spatial_mean_fwd(const float* in, int in_ld, float* out, int out_ld, int N, int H, int W, int C,
                                int freq_only, hipStream_t st);
int tbn_launch_spatial_mean_bwd(const float* dout, int dout_ld, float* din, int din_ld, int N, int H, int W, int C,
                                int freq_only, hipStream_t st);
int tbn_launch_nchw_to_s2d_pad(const float* in, float* out, int N, int C, int H, int W, hipStream_t st);
int tbn_launch_pack_stem_weight_s2d(const float* w, float* wp, int Cout, int C, hipStream_t st);
int tbn_launch_unpack_stem_wgrad_s2d(const float* dwp, float* dw, int Cout, int C, hipStream_t st);
// row-run stem (bordered NHWC image, pool.hip): image [N][HP][WP][C], run length RL = 7*C rounded up to x4, K padded to x32
int tbn_launch_nchw_to_nhwc_pad(const float* in, float* out, int N, int C, int H, int W, int HP, int WP, hipStream_t st);
int tbn_launch_pack_stem_weight_rows(const float* w, float* wp, int Cout, int C, int RL, int K, hipStream_t st);
int tbn_launch_unpack_stem_wgrad_rows(const float* dwp, float* dw, int Cout, int C, int RL, hipStream_t st);
