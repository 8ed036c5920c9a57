/* libtbn_hip.so -- C-ABI of the MI355X (gfx950) TBN hot path.
 *
 * The reference (tridivb/attention_based_tbn) is pure Python on torch.nn and has no native
 * interface; this is the boundary a maintainer binds with ctypes.  Every
 * entry point lists the reference code it replaces.
 *
 * Conventions
 *   - plain C: raw device pointers (fp32 unless noted), explicit sizes / pitches, `void* stream`
 *     is a hipStream_t (pass PyTorch's current stream); no torch types, no allocation, no
 *     synchronisation, no global device state inside -- the caller owns every buffer including
 *     workspaces (sizes from the *_bytes / *_floats queries);
 *   - returns 0 on success, <0 on error (TBN_ERR_*); message via tbn_last_error() (thread-local);
 *     never throws; re-entrant; results are deterministic (no float atomics);
 *   - activations are NHWC with an explicit pitch `ld` (floats between pixels) so operators read
 *     and write channel slices of concat buffers; conv weights are [Cout][R][S][Cin], i.e. the
 *     memory of an OIHW torch parameter in channels_last format.
 */
#ifndef TBN_HIP_H
#define TBN_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define TBN_ERR_ARG (-1)
#define TBN_ERR_LAUNCH (-2)
#define TBN_ERR_UNSUPPORTED (-3)

/* library version in the low 16 bits; bit 16 (0x10000) is set by a -DTBN_EXPERIMENT=1 build, the only build that reads
 * A/B environment knobs (the shipped library reads no environment variable at all) */
int tbn_version(void);
const char* tbn_last_error(void);

/* measurement aid (bench.py roofline): while enabled, every conv-GEMM launch (forward / data-grad /
 * weight-grad implicit GEMMs) is bracketed by hipEvents recorded on the launch stream; entries
 * aggregate, per kernel instantiation, the launch count, summed event time and summed algorithmic
 * FLOPs (2*M*Cout*R*S*Cin of the convolution; padding / zero-insertion work is not counted).
 * tbn_profile_num_entries() synchronises the recorded events.  enable(2) keys entries per layer
 * ("kernel | fwd/dgrad/wgrad <conv name>") instead of per kernel instantiation. */
int tbn_profile_enable(int on);
int tbn_profile_reset(void);
int tbn_profile_num_entries(void);
int tbn_profile_entry(int i, char* name, int name_len, long* launches, double* total_ms, double* total_flops);
/* summed ALGORITHMIC HBM bytes of entry i: per launch the input read once + weights once + output written once (conv /
 * data gradient), dy + x read once + dW written once (weight gradient) -- the denominator of the traffic ratio */
int tbn_profile_entry_bytes(int i, double* total_alg_bytes);
/* measurement aid (bench.py `box.mfma_calibration`): ONE launch of a pure v_mfma_f32_32x32x2_f32 register loop
 * (`workgroups` x 4 waves x `iters` x 16 MFMAs, no memory traffic; sink: any device float, never written) so that a bench
 * line can state what the box it ran on sustains on the instruction the roofline is priced in.  *flops = the MFMA FLOPs of
 * the launch.  Replaces nothing in the reference (which publishes no hardware figures). */
int tbn_diag_mfma_burst(float* sink, int workgroups, int iters, double* flops, void* stream);

/* ---- BN-Inception backbone engine -------------------------------------------------------------
 * replaces: BNInception.features() as instantiated by reference core/models/bn_inception.py:38-107
 * (graph core/models/bn_inception_audio.py:58-404,437-1003 with the 7x7 stem), called from
 * TBNModel.forward core/models/model.py:211-213, and its autograd backward (core/tools/train.py:81).
 * A plan is host-only metadata for one (in_channels, frames, H, W). */
typedef struct tbn_backbone_plan tbn_backbone_plan;

typedef struct {
  char name[64];          /* reference module name, e.g. "inception_3a_1x1" (+"_bn" for its BatchNorm) */
  int cin, cout, ksize, stride, pad;
  size_t weight_offset;   /* floats into the flat weight array: this conv's [cout][k][k][cin] block */
  size_t channel_offset;  /* floats into each flat per-channel array (bias, gamma, beta, mean, var) */
} tbn_conv_info;

typedef struct {
  const float* weight;        /* flat conv weights, tbn_backbone_weight_floats() */
  const float* bias;          /* flat conv biases, tbn_backbone_channel_floats() */
  const float* gamma;         /* BatchNorm weight */
  const float* beta;          /* BatchNorm bias */
  float* running_mean;        /* updated in training mode (momentum) */
  float* running_var;
  float momentum;             /* 0.1 */
  float eps;                  /* 1e-5 */
  void* side_stream;          /* optional second hipStream_t (caller-owned) for tbn_backbone_forward / _backward: BRANCH MODE.
                                 The four branches of an inception block only meet at the concat (reference
                                 core/models/bn_inception_audio.py:437-1003, torch.cat at :485-493 and in every block after);
                                 with a side stream the 3x3 / pool_proj chain of each block runs on it beside the
                                 1x1 -> double_3x3 chain on the launch stream (forked after the 1x1 group's BatchNorm, joined
                                 at the end of the block), so one chain's BN / pool passes sit under the other's GEMMs.
                                 Every layer runs the kernel the serial program would launch for it alone -- sibling-pair
                                 launches (3x3 | double_3x3_1 in one grid) are replaced by the two tuned single launches --
                                 so with pairing off the two programs agree bit for bit (tests/test_branch_gpu.py).  Joined before the
                                 call returns.  NULL = one chain.  Ignored (serial program) while the launch stream is
                                 being captured: forks inside a capture are what aux_stream's note below is about. */
  int flags;                  /* TBN_BACKBONE_RIDERS (1): training passes of the one-chain program put the BN apply / BN-backward
                                 apply of a block's INDEPENDENT column ranges into the grid of a sibling GEMM launch instead of
                                 their own launches -- forward: `1x1` beside `3x3 | double_3x3_1`, `3x3` and `pool_proj` beside
                                 `double_3x3_2`; backward: `1x1`, `3x3`, `pool_proj` beside the data gradient of `double_3x3_2`
                                 (reference dataflow core/models/bn_inception_audio.py:437-1003: these branches only meet at the
                                 concat, :485-493).  Same device code as the stand-alone passes: results are bit-identical with the
                                 flag off (tests/test_riders_gpu.py).  Ignored in branch mode and while the opt-in profiler brackets
                                 the conv launches. */
} tbn_backbone_params;
#define TBN_BACKBONE_RIDERS 1
#define TBN_BACKBONE_STEM_WGRAD_LAST 2   /* one-chain backward without aux stream: the weight gradients of conv2_3x3 and
                                            conv2_3x3_reduce are issued after conv1's pooled BN backward instead of before it
                                            (same kernels, bit-identical results; a scheduling choice between the replicas'
                                            streams) */
#define TBN_BACKBONE_WEIGHTS_FLIPPED 4   /* tbn_backbone_backward only: the flipped / transposed data-gradient weights in the
                                            workspace are current (tbn_backbone_flip_weights ran on this workspace, on the same
                                            stream, after the matching forward and the weights have not changed since): the pass
                                            does not launch the flip itself */

typedef struct {
  float* dweight;             /* same layout as weight; fully overwritten */
  float* dbias;
  float* dgamma;              /* may be NULL */
  float* dbeta;               /* may be NULL */
  int bn_grad_layers;         /* 0 none, 1 first BN only ("partialbn", model.py:164-176), 2 all */
  void* aux_stream;           /* optional second hipStream_t (caller-owned): weight-gradient GEMMs run on it,
                                 overlapping the data-gradient / BN-backward chain; joined before return.  NULL = serial.
                                 Not inside a stream capture: tbn_backbone_backward returns TBN_ERR_UNSUPPORTED when its
                                 launch stream is capturing and aux_stream != NULL.  The fault being fenced off is a NESTED
                                 fork (a stream that itself joined the capture through an event wait forks the aux stream):
                                 hipStreamEndCapture of ROCm 7.x then recurses until the stack overflows.  A single-level
                                 fork from the capture's ORIGIN stream captures fine, but HIP offers no query that tells an
                                 origin stream from a forked one, so the guard refuses that case too -- broader than the
                                 fault, by necessity.  The Python host passes NULL while capturing (serial launches). */
  void (*bucket_cb)(void* user, size_t first_float, size_t num_floats);
                              /* optional (NULL: none).  Gradient buckets for data parallelism (reference: the per-backward
                                 reduce_add_coalesced of nn.DataParallel, core/models/model_builder.py:73-75): called on the
                                 CALLING thread, twice per backward pass, as soon as every launch that writes
                                 dweight[first_float, first_float + num_floats) has been enqueued -- first the weights of
                                 inception_5a..5b, then those of inception_4a..4e (a suffix of the flat tensor each time;
                                 plan order = memory order).  Work the callback enqueues behind the launch stream (an
                                 RCCL all-reduce of that slice) then overlaps the rest of the pass; the remaining prefix
                                 (stem, 3a..3c) is final when the function returns.  The split depends on the graph
                                 only: identical on every replica.  With aux_stream the launch stream is first made to
                                 wait for the weight gradients issued so far.  Not called inside a stream capture. */
  void* bucket_user;
} tbn_backbone_grads;

/* TBN_ERR_UNSUPPORTED for input sizes on which the reference graph itself is inconsistent (its torch.cat of the stride-2
 * branches and the ceil-mode pool of inception_3c / 4e raises) and for frame counts that would make any one tensor of the
 * pass 2 GiB or larger (32-bit byte offsets under the hardware range check: 511 spectrograms of 256x256 or 668 frames of
 * 224x224 per call at most; eval callers chunk the frames, a training step needs a smaller per-GPU batch). */
int tbn_backbone_plan_create(int in_channels, int frames, int height, int width, tbn_backbone_plan** plan);
void tbn_backbone_plan_destroy(tbn_backbone_plan* plan);
int tbn_backbone_num_convs(const tbn_backbone_plan* plan);                       /* 69 */
int tbn_backbone_conv_info(const tbn_backbone_plan* plan, int idx, tbn_conv_info* info);
size_t tbn_backbone_weight_floats(const tbn_backbone_plan* plan);
size_t tbn_backbone_channel_floats(const tbn_backbone_plan* plan);
size_t tbn_backbone_workspace_bytes(const tbn_backbone_plan* plan, int training);
int tbn_backbone_out_shape(const tbn_backbone_plan* plan, int* h, int* w, int* c);
/* 2 when the plan holds a branch-mode program (tbn_backbone_params.side_stream is honoured), else 1 */
int tbn_backbone_num_streams(const tbn_backbone_plan* plan);
/* Opt-in kernel timeline (diagnostics; scripts/step_timeline.py reads the file): while enabled every launch of the library
 * carries a pair of events on its dispatch packet; tbn_timeline_dump synchronises the device and writes one CSV row per
 * launch (Kernel_Name, Queue_Id = stream, Start_Timestamp, End_Timestamp in ns on a common clock).  Unlike an external
 * tracer it leaves the host's launch rate alone, so the overlap of the modality streams is the un-traced step's.
 * enable(1) clears earlier records.  The reference has no counterpart (its only timing: core/tools/train.py:339-351). */
int tbn_timeline_enable(int on);
int tbn_timeline_dump(const char* path);
/* diagnostics: how many conv launches of the plan's LAST training forward / backward pass carried a rider
 * (tbn_backbone_params.flags & TBN_BACKBONE_RIDERS): 2 / 1 per average-pool inception block, 1 / 1 per stride-2 block */
int tbn_backbone_rider_launches(const tbn_backbone_plan* plan, int* forward, int* backward);
/* test / debug aid: location of one conv's tensors inside the workspace (floats). kind 0: z =
 * relu(bn(conv)) destination slice, 1: BN input y (overwritten by dy in backward), 2: gradient
 * wrt z (offset -1 when it is the caller-supplied dfeatures), 3: the conv's whole input buffer, 4: its training-mode
 * BatchNorm coefficients as 4 rows x cout (batch mean | 1/std | scale | shift).  Kind 0 of a conv whose max pool runs
 * inside its BN apply (conv1_7x7_s2, conv2_3x3 in training) fails with TBN_ERR_UNSUPPORTED: that z is never written --
 * it is relu(fma(y, scale, shift)) of kinds 1 and 4. */
int tbn_backbone_tensor_info(const tbn_backbone_plan* plan, const char* conv_name, int kind, long* offset, int* rows,
                             int* cols, int* ld);
/* test / diagnostics aid: the launch choices the plan holds for one conv (after tbn_backbone_autotune: the tuned ones,
 * before: the size heuristics).  Forward choices are kept PER MODE (`training` selects which); the data-gradient ones
 * only exist for training.  out[0..7]  = forward: kernel variant (0 register-staged implicit GEMM, 1 LDS-halo, 2 LDS-DMA,
 * 3 split-K tile), M tile, N tile, LDS stages, issued as a sibling pair (0 / 1; on the pair's first member), pair variant,
 * pair M tile, pair N tile; out[8..15] = the same for the data gradient (zeros when the layer has none / training == 0). */
int tbn_backbone_launch_info(const tbn_backbone_plan* plan, const char* conv_name, int training, int* out16);
/* The tuned launch choices of a plan as a relocatable blob in HOST memory (what tbn_backbone_autotune decides: per GEMM the
 * forward choice per mode, the data-gradient choice, sibling pairing, the weight-gradient tile -- no addresses), so that a
 * plan tuned once can be moved between processes: `DataParallel` broadcasts rank 0's blobs and every replica runs the same
 * kernels, as the reference's nn.DataParallel replicas do by construction (core/models/model_builder.py:73-75,
 * core/models/dataparallel.py:4-6).  export_bytes is a function of the plan's problem only; import validates the header
 * (same in_channels / frames / H / W) and every field against what the launchers accept, and leaves the plan untouched
 * when it refuses.  fingerprint = 64-bit FNV-1a of the blob (bench.py prints it per backbone). */
size_t tbn_backbone_plan_export_bytes(const tbn_backbone_plan* plan);
int tbn_backbone_plan_export(const tbn_backbone_plan* plan, void* buf, size_t bytes);
int tbn_backbone_plan_import(tbn_backbone_plan* plan, const void* buf, size_t bytes);
unsigned long long tbn_backbone_plan_fingerprint(const tbn_backbone_plan* plan);
/* x_nchw: (frames, in_channels, H, W) contiguous, as the reference passes it (model.py:213).
 * training=1: batch-statistics BN, keeps activations in the workspace for backward, updates
 * running stats; training=0: running-stat BN folded into the conv epilogue.
 * *features_out points INTO the workspace: NHWC (frames, h, w, 1024). */
int tbn_backbone_forward(const tbn_backbone_plan* plan, int training, const float* x_nchw,
                         const tbn_backbone_params* params, void* workspace, size_t workspace_bytes,
                         float** features_out, void* stream);
/* optional one-time tuning of the per-layer GEMM tiles on the real shapes (times each candidate with
 * hipEvents).  The ONLY entry point that synchronises the stream; clobbers the activations held in
 * `workspace`, so call it between steps (e.g. right after the first forward of a new plan). */
int tbn_backbone_autotune(tbn_backbone_plan* plan, int training, const tbn_backbone_params* params, void* workspace,
                          size_t workspace_bytes, void* stream);
/* dfeatures: NHWC (frames, h, w, 1024) gradient of *features_out; needs the workspace of the
 * matching training forward untouched. */
int tbn_backbone_backward(const tbn_backbone_plan* plan, const float* dfeatures,
                          const tbn_backbone_params* params, const tbn_backbone_grads* grads, void* workspace,
                          size_t workspace_bytes, void* stream);
/* The first launch of tbn_backbone_backward on its own: the flipped / transposed copy of every data-gradient weight
 * (what autograd's conv backward gets from cuDNN implicitly; reference: loss.backward(), core/tools/train.py:86) into the
 * training workspace.  It depends on the weights only, so a caller with several backbones on several streams can issue it on
 * each backbone's stream right after the streams were joined for the heads -- it then runs beside the heads' small kernels
 * instead of between them and the first backward GEMM -- and pass TBN_BACKBONE_WEIGHTS_FLIPPED to the backward pass. */
int tbn_backbone_flip_weights(const tbn_backbone_plan* plan, const tbn_backbone_params* params, void* workspace,
                              size_t workspace_bytes, void* stream);

/* ---- single operators (same kernels the engine uses; used by the heads and the parity tests) ---- */

/* nn.Conv2d forward, NHWC.  epilogue: 0 = +bias (optional ReLU / accumulate via flags: 1 accumulate
 * into out, 2 ReLU); 1 = bias-FREE conv output plus per-channel (sum, sumsq) partials for a following
 * training-mode BN (stat_partial[tbn_conv2d_stat_tiles()][2][cout]) -- a per-channel constant cancels in
 * batch-stat BN, so `bias` is ignored here and only enters the running mean (tbn_backbone_* does that);
 * 2 = relu(conv * scale + shift) for eval BN (`bias` ignored: fold it into shift = beta + (bias-mean)*scale).
 * replaces: each nn.Conv2d of bn_inception_audio.py:24-401 (cuDNN).  cin must be a multiple of 32, ksize <= 3 (the
 * 7x7 stem exists only inside tbn_backbone_*: it reads a space-to-depth image the engine lays out itself).
 * flags also selects the kernel variant (test / tuning aid; 0 = generic): 4 = LDS-halo (3x3 / stride 1 / pad 1 only),
 * 8 = LDS-DMA staging, 16 = 32-row tiles whose four waves split K (statistics partial rows are then per 32*mt output
 * rows, tiles mt, nt in {1,2}).  Without a variant flag, epilogue-0 / -2 launches of at most 128 tiles and k*k*cin >= 256
 * (the head Linear layers, M = 96 rows) take the split-K tile kernel by themselves; every variant is deterministic. */
int tbn_conv2d_fwd(const float* in, int in_ld, const float* weight, const float* bias, float* out, int out_ld,
                   int n, int h, int w, int cin, int cout, int ksize, int stride, int pad, int epilogue, int flags,
                   const float* scale, const float* shift, float* stat_partial, void* stream);
int tbn_conv2d_stat_tiles(int n, int h, int w, int cin, int cout, int ksize, int stride, int pad);
/* tuning / test aid: as tbn_conv2d_fwd (epilogue 0 or 1) with an explicit tile: the workgroup computes
 * (128*mt) x (32*nt) outputs, mt in {1,2}, nt in {1..4}; (0,0) = built-in heuristic */
int tbn_conv2d_fwd_tile(const float* in, int in_ld, const float* weight, const float* bias, float* out, int out_ld,
                        int n, int h, int w, int cin, int cout, int ksize, int stride, int pad, int epilogue, int flags,
                        float* stat_partial, int mt, int nt, void* stream);
/* data gradient: din (n,h,w,cin) = conv_transpose(dout).  workspace: cout*k*k*cin floats. */
int tbn_conv2d_dgrad(const float* dout, int dout_ld, const float* weight, float* din, int din_ld, int n, int h, int w,
                     int cin, int cout, int ksize, int stride, int pad, int accumulate, float* workspace,
                     void* stream);
/* ---- one convolution launch in full detail (test / tuning aid) --------------------------------
 * Reaches every kernel variant, tile, epilogue and the fused BN-backward reduce that the backbone engine's autotuner
 * can put on a layer -- including the forms the plain entry points above never select by themselves: explicit tiles of
 * the parity-phase launch of a strided data gradient, data-gradient / eval epilogues of the LDS-halo / LDS-DMA /
 * split-K tile kernels, the reduce epilogue, and two sibling convolutions in ONE launch.
 * replaces: the same nn.Conv2d forward / backward of bn_inception_audio.py:24-401 as tbn_conv2d_*. */
typedef struct {
  const float* y;         /* BN input of the producer layer at its column 0 (pitch y_ld), output pixel order */
  int y_ld;
  int col_begin;          /* first output column of this layer (multiple of 32, ascending, disjoint) */
  int channels;
  int stat_offset;        /* this layer's channel 0 inside the red_stats arrays */
  float* partial;         /* [tbn_conv_partial_rows()][2][channels]: per M tile  sum g | sum g * xhat */
} tbn_conv_red;
typedef struct {
  const float* in;        /* forward: input (n,h,w,cin); data gradient: dout (n,oh,ow,cout) */
  int in_ld;
  const float* weight;    /* [cout][k][k][cin] */
  const float* bias;      /* epilogue 0 of a forward launch; may be NULL */
  float* out;             /* forward: (n,oh,ow,cout); data gradient: din (n,h,w,cin) */
  int out_ld;
  int n, h, w, cin, cout, ksize, stride, pad;   /* geometry of the FORWARD convolution */
  int dgrad;              /* 0: forward, 1: data gradient of that convolution (workspace: cout*k*k*cin floats) */
  int epilogue;           /* forward: 0 / 1 / 2 as tbn_conv2d_fwd; data gradient: 0 */
  int flags;              /* 1 accumulate, 2 ReLU; kernel variant 4 LDS-halo, 8 LDS-DMA, 16 split-K tile (0 generic) */
  int stages;             /* generic kernel: LDS stages 1 / 2 (0 = default) */
  const float* scale;     /* epilogue 2 */
  const float* shift;
  float* stat_partial;    /* epilogue 1: [tbn_conv_partial_rows()][2][cout] */
  int nred;               /* data gradient: fused BN-backward reduce over nred producer layers (0 = off) */
  tbn_conv_red red[4];
  const float* red_stats; /* mean | rstd | scale | shift, each red_stats_stride floats */
  int red_stats_stride;
} tbn_conv_desc;
/* rows of stat_partial / red[i].partial that a launch with M tile `mt` writes (pair = 1: issued by tbn_conv_launch_pair) */
int tbn_conv_partial_rows(const tbn_conv_desc* d, int mt, int pair);
int tbn_conv_launch(const tbn_conv_desc* d, int mt, int nt, float* workspace, void* stream);
/* two independent unit-stride convolutions with the same epilogue in ONE launch (the 3x3 | double_3x3_1 siblings of an
 * inception block): variant 0 LDS-halo (both 3x3 / stride 1 / pad 1), 1 / 2 generic kernel with 1 / 2 LDS stages;
 * mt, nt in {1,2} */
int tbn_conv_launch_pair(const tbn_conv_desc* a, const tbn_conv_desc* b, int variant, int mt, int nt,
                         float* workspace_a, float* workspace_b, void* stream);

/* weight gradient: dweight [cout][k][k][cin].  workspace: tbn_conv2d_wgrad_workspace_floats(). */
size_t tbn_conv2d_wgrad_workspace_floats(int n, int h, int w, int cin, int cout, int ksize, int stride, int pad);
int tbn_conv2d_wgrad(const float* dout, int dout_ld, const float* in, int in_ld, float* dweight, int n, int h, int w,
                     int cin, int cout, int ksize, int stride, int pad, float* workspace, void* stream);

/* nn.Linear / nn.Conv1d(k=1) (model.py:337-386 Fusion/Classifier, model.py:62-67 pe.1, the MHA
 * projections attention.py:48-57): out[m][n] = x[m][:] . w[n][:] + bias[n].  k % 32 == 0. */
int tbn_linear_fwd(const float* x, int x_ld, const float* w, const float* bias, float* out, int out_ld, int m, int k,
                   int n, int relu, void* stream);
/* dx[m][k] (+)= dy[m][:] . w[:][k]; n % 32 == 0; workspace n*k floats */
int tbn_linear_dgrad(const float* dy, int dy_ld, const float* w, float* dx, int dx_ld, int m, int k, int n,
                     int accumulate, float* workspace, void* stream);
/* dw[n][k] = dy^T x ; dbias[n] = column sums of dy (dbias may be NULL); n % 4 == 0, k % 4 == 0 */
size_t tbn_linear_wgrad_workspace_floats(int m, int k, int n);
int tbn_linear_wgrad(const float* dy, int dy_ld, const float* x, int x_ld, float* dw, float* dbias, int m, int k,
                     int n, float* workspace, void* stream);

/* training-mode BatchNorm2d + ReLU on NHWC (p pixels x c channels), standalone form.
 * replaces nn.BatchNorm2d(...).train() + ReLU(inplace).  Workspace floats: tbn_bn_workspace_floats(). */
size_t tbn_bn_workspace_floats(int p, int c);
int tbn_bn_relu_train_fwd(const float* y, int p, int c, const float* gamma, const float* beta, float* running_mean,
                          float* running_var, float momentum, float eps, float* save_mean, float* save_rstd,
                          float* scale, float* shift, float* z, int z_ld, float* workspace, void* stream);
int tbn_bn_relu_train_bwd(const float* dz, int dz_ld, const float* y, int p, int c, const float* save_mean,
                          const float* save_rstd, const float* scale, const float* shift, float* dy, float* dgamma,
                          float* dbeta, float* workspace, void* stream);

/* training-mode BatchNorm2d + ReLU + MaxPool2d(3, stride, pad, ceil_mode) in one pass over y: z = relu(bn(y)) is pooled in
 * registers and never written (what the backbone does for conv1_7x7_s2 -> pool1 and conv2_3x3 -> pool2, the two largest
 * tensors of the network); the backward gathers dz from (dpooled, argmax) on the fly.  y: (n,h,w,c) NHWC, pooled:
 * (n,oh,ow,c) with pitch pooled_ld, argmax: n*oh*ow*c bytes.  Workspace: tbn_bn_workspace_floats(n*h*w, c).
 * replaces: BatchNorm2d.train() + ReLU + MaxPool2d of bn_inception_audio.py:21-23,28-34. */
int tbn_bn_relu_maxpool_train_fwd(const float* y, int n, int h, int w, int c, const float* gamma, const float* beta,
                                  float* running_mean, float* running_var, float momentum, float eps, float* save_mean,
                                  float* save_rstd, float* scale, float* shift, float* pooled, int pooled_ld,
                                  uint8_t* argmax, int oh, int ow, int stride, int pad, float* workspace, void* stream);
int tbn_bn_relu_maxpool_train_bwd(const float* dpooled, int dpooled_ld, const uint8_t* argmax, const float* y, int n, int h,
                                  int w, int c, int oh, int ow, int stride, int pad, const float* save_mean,
                                  const float* save_rstd, const float* scale, const float* shift, float* dy,
                                  float* dgamma, float* dbeta, float* workspace, void* stream);

/* pooling (bn_inception_audio.py:21-23,86-88,155-157,394-396; ceil_mode handled by oh/ow) */
int tbn_maxpool3_fwd(const float* in, int in_ld, float* out, int out_ld, uint8_t* argmax, int n, int h, int w, int c,
                     int oh, int ow, int stride, int pad, void* stream);
int tbn_maxpool3_bwd(const float* dout, int dout_ld, const uint8_t* argmax, float* din, int din_ld, int n, int h,
                     int w, int c, int oh, int ow, int stride, int pad, int accumulate, void* stream);
int tbn_avgpool3_fwd(const float* in, int in_ld, float* out, int out_ld, int n, int h, int w, int c, int accumulate,
                     void* stream); /* 3x3 s1 p1 count_include_pad; self-adjoint => also its backward */
/* BNInception.logits override, bn_inception.py:16-35: freq_only=1 -> mean over H only (attended
 * audio, out (n,w,c)); else global mean (out (n,c)) */
int tbn_spatial_mean_fwd(const float* in, int in_ld, float* out, int out_ld, int n, int h, int w, int c,
                         int freq_only, void* stream);
int tbn_spatial_mean_bwd(const float* dout, int dout_ld, float* din, int din_ld, int n, int h, int w, int c,
                         int freq_only, void* stream);

/* ---- mid-level fusion heads ---------------------------------------------------------------- */
/* PositionalEncoding "concat" (attention.py:8-45): out[r][t][0:c]=feat, [c:c+pe_dim]=pe[:,t], zero
 * padded to out_ld. pe is (pe_dim, t) row-major. */
int tbn_pe_concat_fwd(const float* feat, int feat_ld, const float* pe, float* out, int out_ld, int r, int t, int c,
                      int pe_dim, void* stream);
/* nn.GroupNorm(groups, c) over (r, t, c) rows (model.py:62-67 pe.2) */
int tbn_groupnorm_fwd(const float* x, float* y, const float* gamma, const float* beta, float* save_mean,
                      float* save_rstd, int r, int t, int c, int groups, float eps, void* stream);
/* dgamma_part / dbeta_part: (r, c) per-sample partials, reduce with tbn_colsum */
int tbn_groupnorm_bwd(const float* dy, const float* x, const float* gamma, const float* save_mean,
                      const float* save_rstd, float* dx, float* dgamma_part, float* dbeta_part, int r, int t, int c,
                      int groups, void* stream);
int tbn_colsum(const float* x, int x_ld, float* out, int rows, int cols, void* stream);
/* torch.nn.MultiheadAttention core for L_q = 1 (attention.py:48-57, model.py:231-237), one wave per
 * (sample, head): q (r,e) projected query (unscaled; scores = scale * q.k, scale = head_dim^-0.5);
 * kv (r,t,2e) = [k | v] projections.  drop_mask (r,heads,t) holds 0 or 1/(1-p), NULL in eval.
 * Outputs: ctx (r,e) (input of out_proj); probs, 2*r*heads*t floats = pre-dropout softmax (saved
 * for backward) followed by the post-dropout weights; avg_w (r,t) head-mean of the post-dropout
 * weights (what nn.MultiheadAttention returns). */
int tbn_mha_q1_fwd(const float* q, const float* kv, const float* drop_mask, float* ctx, float* probs, float* avg_w,
                   int r, int t, int e, int heads, float scale, void* stream);
int tbn_mha_q1_bwd(const float* dctx, const float* davg_w, const float* q, const float* kv, const float* probs,
                   const float* drop_mask, float* dq, float* dkv, int r, int t, int e, int heads, float scale,
                   void* stream);
/* fixed attention (model.py:224-228): out[r][c] = sum_t feat[r][t][c] * w[r][t] */
int tbn_weighted_sum_fwd(const float* feat, const float* w, float* out, int out_ld, int r, int t, int c, void* stream);
int tbn_weighted_sum_bwd(const float* dout, int dout_ld, const float* w, float* dfeat, int r, int t, int c,
                         void* stream);
/* temporal consensus (model.py:178-203): out[b][c] = mean_n x[b*n+i][c] */
int tbn_segment_mean_fwd(const float* x, float* out, int b, int n, int c, void* stream);
int tbn_segment_mean_bwd(const float* dout, float* dx, int b, int n, int c, void* stream);
/* nn.Dropout(p) of the fusion layer (reference model.py:352-362) from a caller-drawn uniform tensor `rnd` in [0, 1):
 * mask = rnd >= p ? 1 / (1 - p) : 0, y = x * mask, both written (the mask is what backward multiplies by: tbn_mul_mask) */
int tbn_dropout_fwd(const float* x, const float* rnd, float p, float* y, float* mask, size_t count, void* stream);
/* The cross-entropy losses of up to 4 classification heads that share ONE score matrix (reference model.py:272-279: one
 * nn.CrossEntropyLoss(mean) per class key, `verb` and `noun`): head h owns columns [col0[h], col0[h] + ncls[h]) of
 * scores (batch, ld), labels[h] = int64[batch] device pointers (host array of them).  A label of -100 (the criterion's
 * default ignore_index) takes its row out of loss, gradient and mean; any other label outside [0, ncls[h]) gives NaN.
 * Writes loss[h] (device, mean over the rows not ignored, fixed summation order), rowloss (scratch, num_heads * batch floats) and dscores (batch, ld) =
 * d(sum_h loss[h]) / d(scores) for the heads' columns (other columns untouched).  tbn_ce_heads_bwd scales a head's columns
 * of dscores by upstream[h] (device) into `out` (may alias dscores): the backward for arbitrary per-head weights. */
int tbn_ce_heads_fwd(const float* scores, int ld, int batch, int num_heads, const int* col0, const int* ncls,
                     const long long* const* labels, float* rowloss, float* loss, float* dscores, void* stream);
int tbn_ce_heads_bwd(const float* dscores, int ld, int batch, int num_heads, const int* col0, const int* ncls,
                     const float* upstream, float* out, void* stream);
/* y = x * mask (dropout with a caller-generated keep/scale mask); in place allowed */
int tbn_mul_mask(const float* x, const float* mask, float* y, size_t count, void* stream);
/* dx = dy * [y > 0] (* mask if non-NULL): backward of Linear->ReLU->Dropout (model.py:337-362) */
int tbn_relu_mask_bwd(const float* dy, const float* y, const float* mask, float* dx, size_t count, void* stream);

/* ---- spectrogram --------------------------------------------------------------------------- */
/* log-power STFT of reference core/dataset/dataset.py:461-495 (librosa.stft n_fft=511, hop=120,
 * win=240 hann, centre, zero pad): wave (nseg, len) -> spec (nseg, 256, 1+(len-1)/120).
 * twiddle: tbn_stft_twiddle_floats() floats filled by tbn_stft_make_twiddle() (host fp64 -> fp32). */
size_t tbn_stft_twiddle_floats(void);
int tbn_stft_make_twiddle(float* host_buffer);
int tbn_stft_logpower(const float* wave, int nseg, int len, const float* twiddle, float* spec, float eps,
                      void* stream);

/* ---- train-step shell and metrics (the steps right after the path) -- */
/* Multi-tensor clip_grad_norm_ + SGD(momentum) of reference core/tools/train.py:82-94,190-202
 * (torch.nn.utils.clip_grad_norm_ norm_type 2; torch.optim.SGD dampening 0, no nesterov).  All tensors of a
 * call ride in one launch (16-byte aligned tensors move 16 B per lane, others 4 B); `momentum` buffers start as zeros (first step
 * m = d, as torch's clone).  Nothing is read back: the clip coefficient stays on the device. */
#define TBN_OPT_MAX_TENSORS 48
typedef struct tbn_opt_tensor {
  void* param;        /* fp32 parameter (may be NULL for the norm / scale entries) */
  const void* grad;   /* fp32 gradient */
  void* momentum;     /* fp32 momentum buffer (NULL when momentum == 0) */
  size_t count;       /* elements */
} tbn_opt_tensor;
/* number of per-workgroup partial sums tbn_opt_sqnorm_partials writes for these tensors */
int tbn_opt_num_partials(const tbn_opt_tensor* tensors, int num_tensors);
/* partials[i] = sum of squares of one 16384-element chunk of the gradients */
int tbn_opt_sqnorm_partials(const tbn_opt_tensor* tensors, int num_tensors, float* partials, void* stream);
/* total_norm[0] = sqrt(sum partials) (fp64, fixed order); coef[0] = min(1, max_norm / (total_norm + 1e-6)) */
int tbn_opt_clip_coef(const float* partials, int num_partials, float max_norm, float* total_norm, float* coef,
                      void* stream);
/* grad *= coef[0] in place (what clip_grad_norm_ leaves behind; skipped when coef == 1) */
int tbn_opt_scale_grads(const tbn_opt_tensor* tensors, int num_tensors, const float* coef, void* stream);
/* d = grad * grad_scale[0] (1 if NULL) + weight_decay * p ; m = momentum * m + d ; p -= lr * m */
int tbn_opt_sgd_step(const tbn_opt_tensor* tensors, int num_tensors, float lr, float momentum, float weight_decay,
                     const float* grad_scale, void* stream);
/* Metric._get_correct_score (core/utils/metric.py:137-157): scores (batch, classes) pitch scores_ld, target int64.
 * correct (k, batch) uint8 = [j-th ranked class == target]; pred (k, batch) int64 ranked classes (may be NULL);
 * conf_mat (classes, classes) float, conf_mat[target][top1] += 1 (may be NULL).  Ties rank the lower index first. */
int tbn_topk_correct(const float* scores, int scores_ld, const long long* target, int batch, int classes, int k,
                     unsigned char* correct, long long* pred, float* conf_mat, void* stream);

/* ---- on-device visual input pipeline (the step right before the path) ---- */
/* MultiScaleCrop / Rescale (cv2.resize INTER_LINEAR, 8-bit fixed point) -> CenterCrop / crop -> RandomHorizontalFlip
 * -> Stack -> ToTensor (/255) -> Normalize of reference core/dataset/transform.py:9-543 as composed by
 * core/utils/create_dataloader.py:19-81, in one pass over the pixels.  frames: (n_img, height, width, channels)
 * uint8 on the device.  The source box (box_*) is resized to resized_w x resized_h (equal sizes: no interpolation),
 * the window (crop_x, crop_y, out_w, out_h) of that is taken, flipped horizontally if `flip`; `stack` consecutive
 * single/multi-channel images form the channels of one sample; out: (n_img/stack, channels*stack, out_h, out_w)
 * fp32 = ((v [/255]) - mean[c % n_stat]) / std[c % n_stat] (n_stat = 0: no normalisation).  The random choices
 * (crop box, flip) are the caller's: same host RNG draws as the reference. */
int tbn_frames_to_tensor(const unsigned char* frames, int n_img, int height, int width, int channels, int box_x,
                         int box_y, int box_w, int box_h, int resized_w, int resized_h, int crop_x, int crop_y,
                         int out_w, int out_h, int flip, int stack, const float* mean, const float* std_dev,
                         int n_stat, int div255, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TBN_HIP_H */
