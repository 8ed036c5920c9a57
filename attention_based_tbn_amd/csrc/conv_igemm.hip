// Implicit-GEMM 2-D convolution for gfx950 on the fp32 MFMA (v_mfma_f32_32x32x2_f32).
//
// Replaces, for the TBN hot path, what the reference gets from cuDNN through nn.Conv2d
// (graph: reference core/models/bn_inception_audio.py:24-401; 7x7 stem bn_inception.py:75-77).
//
// Data layout (MI355X-first, not the reference's NCHW):
//   activations NHWC fp32 with an explicit row pitch `ld` so a conv can read / write a channel
//   slice of a wider concat buffer; weights [Cout][R][S][Cin] (= torch channels_last memory of the
//   OIHW parameter), i.e. GEMM-B rows are K-contiguous.
// GEMM view:  M = N*OH*OW output pixels, N = Cout, K = R*S*Cin.
//   A (pixels x K) is gathered on the fly (im2col never materialised), staged through LDS in
//   [row][32+4] fp32 tiles: 128-B coalesced global reads per 8 lanes, ds_write_b128, and
//   conflict-free ds_read_b128 fragments (k is permuted identically for A and B, so one b128
//   read feeds 4 MFMAs).  4 waves per workgroup stacked along M, each owning (32*MT) x (32*NT).
//   blockIdx -> tile mapping is XCD-aware: the 8 tiles that share an activation row-panel
//   sit on one XCD so the panel is fetched into one L2 only.
// Epilogues (runtime switch): plain(+bias,+relu,+accumulate) | raw+bias with fused per-channel
//   sum / sum-of-squares partials for training-mode BatchNorm | eval BN folded scale/shift + ReLU.
// The same kernel is the data-gradient: dgrad = conv of dy with tap-flipped, channel-transposed
//   weights (`up` = forward stride handles strided layers by zero-insertion on the fly).
// ROWMODE is the 7x7 / stride 2 stem as a 4x4 / stride 1 conv on the 2x2 space-to-depth image (pool.hip): the K of an
//   output pixel is R contiguous runs of `Cin` floats (4 s2d pixels x 4*cin channels), one per s2d row, packed back to
//   back -- 32-float K chunks straddle runs -- and the image carries its zero border physically, so the loads need
//   no masks at all.
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include <cstring>
#include "tbn_common.h"
#include "tbn_kernels.h"
#include "tbn_rider_dev.h"

#define LDT 36  // LDS row pitch in floats (32 + 4): 16-B aligned rows, conflict-free b128 reads
// Timing ablations of the main loop (scripts/conv_ablate.py) exist only in a -DTBN_ABLATE=1 build: as run-time
// flags their scalar branches cut the K loop into basic blocks and cost the production kernel ~10 %.
#ifndef TBN_ABLATE
#define TBN_ABLATE 0
#endif
#ifndef TBN_DIAG
#define TBN_DIAG 0   // timing-diagnostic build: see ConvP::fold_scale
#endif
#define ABL(bit) (TBN_ABLATE && (p.flags & (bit)))
#define WABL(bit) (TBN_ABLATE && (p.ablate & (bit)))   // weight-gradient kernel: 1 no loop loads, 2 no LDS stores, 4 no MFMA

// Hardware-bounds-checked 16-B loads.  ROCm 7.2's clang lowers __builtin_amdgcn_raw_buffer_load_b128 to a
// ONE-dword load, so the LLVM intrinsic is bound directly (same idiom as composable_kernel).
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ f32x4 tbn_llvm_buffer_load_f32x4(i32x4 srsrc, int voffset, int soffset, int aux) __asm(
    "llvm.amdgcn.raw.buffer.load.v4f32");
#define TBN_OOB 0x80000000u  // byte offset beyond any buffer extent (< 2 GiB enforced) -> hardware returns 0

// 128-bit buffer descriptor from wave-uniform kernel arguments (base, extent in bytes)
__device__ __forceinline__ i32x4 make_rsrc(const void* ptr, unsigned bytes) {
  union {
    i32x4 v;
    struct {
      const void* p;
      unsigned range, cfg;
    } s;
  } u;
  u.s.p = ptr;
  u.s.range = bytes;
  u.s.cfg = 0x00020000u;
  i32x4 r;
  r.x = __builtin_amdgcn_readfirstlane(u.v.x);
  r.y = __builtin_amdgcn_readfirstlane(u.v.y);
  r.z = __builtin_amdgcn_readfirstlane(u.v.z);
  r.w = __builtin_amdgcn_readfirstlane(u.v.w);
  return r;
}

__device__ void tbn_llvm_buffer_store_f32(float data, i32x4 srsrc, int voffset, int soffset, int aux) __asm(
    "llvm.amdgcn.raw.buffer.store.f32");
__device__ float tbn_llvm_buffer_load_f32(i32x4 srsrc, int voffset, int soffset, int aux) __asm(
    "llvm.amdgcn.raw.buffer.load.f32");

__device__ __forceinline__ float4 buf_load4(i32x4 r, unsigned voff, unsigned soff = 0u) {
  const f32x4 v = tbn_llvm_buffer_load_f32x4(r, (int)voff, (int)soff, 0);
  return make_float4(v.x, v.y, v.z, v.w);
}

// ---------------------------------------------------------------- epilogue (shared by the GEMM bodies)
// EPI 1 / 2 work on the bias-free accumulator (training: a per-channel constant cancels in the batch-stat
// BN; eval: the bias is folded into `shift`).  Rows >= M and columns >= Cout hold exact zeros, so the
// statistics need no masking.  Stores are buffer stores: the lane offset is computed once per 32x32
// sub-tile, the per-register row step rides in the scalar offset -> no VALU address math, no branches.
// Precondition: every wave has passed a barrier after its last LDS tile read (`lds` is reused for the partial sums).
// WM = waves stacked along M: 4 (the GEMM bodies whose waves own 32*MT rows each) or 1 (conv_sk4_body: ONE wave holds the
// finished (32*MT) x (32*NT) tile, the partial sums go straight to global memory, no barrier).
template <int MT, int NT, int EPI, bool RED, int WM = 4>
__device__ __forceinline__ void conv_epilogue(const ConvP& p, f32x16 (&acc)[MT][NT], float* lds, const int tm,
                                              const int m0, const int n0) {
  constexpr int BM = 32 * WM * MT, BN = 32 * NT;
  const int tid = threadIdx.x, lane = tid & 63, wave = WM == 1 ? 0 : (tid >> 6);
  const int lrow = lane & 31, lhalf = lane >> 5;
  float* red = lds;  // [2][4 waves][BN] for the BN-statistics partials (tiles are dead now)
  const int mrow0 = m0 + wave * 32 * MT + 4 * lhalf;  // + i*32 + 8*g + q  (accumulator register e = 4*g + q)
  const bool tile_full = (m0 + BM <= p.M);
  constexpr bool scatter = (EPI == 3);  // strided data-gradient phase
  constexpr bool SUMS = (EPI == 1) || RED;   // (the K loop ends with a barrier: `red` may overlay the tiles)
  // VALU diet of the epilogue (round 3): the static instruction count of a <1,1> data-gradient tile with the reduce was
  // ~650 VALU (~450 of them here) against 16 MFMAs per K-step -- 2.2 VALU per MFMA over an 18-step tile, 6.8 over the
  // 6 steps of the short-K 1x1 groups -- because block-uniform run-time cases (ragged last M tile, accumulate, ReLU,
  // bias) were evaluated per element with selects.  A full tile without those takes the LEAN loop: per element one
  // store (+ 2 VALU for the statistics, + 6 for the reduce); everything else keeps the general loop.
  // (round 4: an accumulating data gradient -- the 1x1 groups of 3c / 4e / 5b, whose block input also receives a max
  //  pool's gradient -- takes the lean loop too, with its 16 old values loaded up front like the reduce's y values: these
  //  were the slowest data gradients of a backbone, 72-92 TF/s, on the ~450-VALU general loop because of that one flag)
  const bool lean = tile_full && !scatter && (EPI != 0 || ((p.flags & CONV_FLAG_RELU) == 0 && p.bias == nullptr));
  const bool accum = (EPI == 0) && (p.flags & CONV_FLAG_ACCUM) != 0;   // block-uniform
  // strided data-gradient phase: output pixel of each tile row, decoded ONCE per row into LDS (was: two magic-number
  // divisions per ELEMENT, ~25 VALU x 16 elements x NT sub-tiles per lane)
  unsigned* opix_tab = reinterpret_cast<unsigned*>(lds + 2 * 4 * BN);
  if (scatter) {
    if (tid < BM) {
      const int m = m0 + tid;
      const uint32_t n = fdiv((uint32_t)m, p.div_ohw);
      const uint32_t rem = (uint32_t)m - n * p.div_ohw.d;
      const uint32_t a = fdiv(rem, p.div_ow);
      const uint32_t b = rem - a * p.div_ow.d;
      const unsigned opix = (unsigned)(((int)n * p.OH + ((int)a * p.out_sy + p.out_oy)) * p.OW + ((int)b * p.out_sx + p.out_ox));
      opix_tab[tid] = m < p.M ? opix : 0xffffffffu;
    }
    __syncthreads();
  }
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int colb = n0 + j * 32;
    if (colb >= p.Cout) continue;  // block-uniform
    const int col = colb + lrow;
    const bool col_ok = col < p.Cout;
    int sg = 0;
    if (p.nseg > 1 && colb >= p.seg[1].col_begin) sg = 1;
    if (p.nseg > 2 && colb >= p.seg[2].col_begin) sg = 2;
    if (p.nseg > 3 && colb >= p.seg[3].col_begin) sg = 3;
    const int old = p.seg[sg].ld;
    const i32x4 o_rsrc = make_rsrc(p.seg[sg].ptr, p.seg_bytes[sg]);
    const unsigned col_off = (unsigned)(col - p.seg[sg].col_begin) * 4u;
    const float bias = (EPI == 0 || EPI == 3) ? ((p.bias != nullptr && col_ok) ? p.bias[col] : 0.f) : 0.f;
    float sc = 1.f, sh = 0.f;
    bool raw = false;
    if (EPI == 2) {
      raw = (p.raw_seg1 == sg + 1);   // block-uniform
      if (col_ok && !raw) {
        sc = p.scale[col];
        sh = p.shift[col];
      }
    }
    // fused BN-backward reduce: the producer layer of these 32 columns (block-uniform, segments start on x32 columns)
    int rs = 0, lc = 0;
    bool red_on = false;
    i32x4 y_rsrc = o_rsrc;
    unsigned ycol_off = 0u;
    int yld = 0;
    float b_sc = 0.f, b_sh = 0.f, b_mu = 0.f, b_rs = 0.f;
    if (RED) {
      if (p.nred > 1 && colb >= p.red[1].col_begin) rs = 1;
      if (p.nred > 2 && colb >= p.red[2].col_begin) rs = 2;
      if (p.nred > 3 && colb >= p.red[3].col_begin) rs = 3;
      red_on = p.red[rs].y != nullptr && colb < p.red[rs].col_begin + p.red[rs].C;
      if (red_on) {
        y_rsrc = make_rsrc(p.red[rs].y, p.red[rs].y_bytes);
        yld = p.red[rs].y_ld;
        lc = col - p.red[rs].col_begin;
        ycol_off = (unsigned)lc * 4u;
        if (lc < p.red[rs].C) {
          const float* stp = p.red[rs].stats + p.red[rs].c_off + lc;
          b_mu = stp[0];
          b_rs = stp[p.red_chan];
          b_sc = stp[2 * p.red_chan];
          b_sh = stp[3 * p.red_chan];
        }
      }
    }
    float s1 = 0.f, s2 = 0.f;
    if (lean) {
      // columns without a BN layer behind them (or beyond it) keep b_sc = b_sh = 0: fma(y, 0, 0) > 0 is false -> g = 0
      const float b_nmr = -b_mu * b_rs;       // xhat = fma(y, rstd, -mean * rstd)
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const unsigned vbase = col_ok ? (unsigned)(mrow0 + i * 32) * (unsigned)old * 4u + col_off : TBN_OOB;
        const unsigned ybase = (RED && red_on && col_ok) ? (unsigned)(mrow0 + i * 32) * (unsigned)yld * 4u + ycol_off : TBN_OOB;
        float yv[16], ov[16];
        if (EPI == 0 && accum) {
#pragma unroll
          for (int e = 0; e < 16; ++e)
            ov[e] = tbn_llvm_buffer_load_f32(o_rsrc, (int)vbase, (int)((unsigned)((8 * (e >> 2) + (e & 3)) * old) * 4u), 0);
        }
        if (RED) {
#pragma unroll
          for (int e = 0; e < 16; ++e)
            yv[e] = tbn_llvm_buffer_load_f32(y_rsrc, (int)ybase, (int)((unsigned)((8 * (e >> 2) + (e & 3)) * yld) * 4u), 0);
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int dm = 8 * (e >> 2) + (e & 3);
          float v = acc[i][j][e];
          if (EPI == 0 && accum) v += ov[e];
          if (EPI == 1) {
            s1 += v;
            s2 = fmaf(v, v, s2);
          } else if (EPI == 2) {
            if (!raw) v = fmaxf(fmaf(v, sc, sh), 0.f);
          }
          tbn_llvm_buffer_store_f32(v, o_rsrc, (int)vbase, (int)((unsigned)(dm * old) * 4u), 0);
          if (RED) {
            const float g = fmaf(yv[e], b_sc, b_sh) > 0.f ? v : 0.f;
            s1 += g;
            s2 = fmaf(g, fmaf(yv[e], b_rs, b_nmr), s2);
          }
        }
      }
    } else {
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const unsigned vbase = col_ok ? (unsigned)(mrow0 + i * 32) * (unsigned)old * 4u + col_off : TBN_OOB;
      const unsigned ybase = (RED && red_on && col_ok) ? (unsigned)(mrow0 + i * 32) * (unsigned)yld * 4u + ycol_off : TBN_OOB;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int dm = 8 * (e >> 2) + (e & 3);
        float v = acc[i][j][e];
        if (EPI == 1) {
          s1 += v;
          s2 = fmaf(v, v, s2);
        } else if (EPI == 2) {
          if (!raw) v = fmaxf(fmaf(v, sc, sh), 0.f);
        } else {
          v += bias;
        }
        unsigned voff = vbase, soff = (unsigned)(dm * old) * 4u;
        unsigned yvoff = ybase, ysoff = (unsigned)(dm * yld) * 4u;
        if (scatter) {
          const unsigned opix = opix_tab[wave * 32 * MT + i * 32 + 4 * lhalf + dm];   // 0xffffffff: row >= M
          const bool ok = opix != 0xffffffffu && col_ok;
          voff = ok ? __umul24(opix, (unsigned)old * 4u) + col_off : TBN_OOB;
          soff = 0u;
          if (RED) {
            yvoff = (ok && red_on) ? __umul24(opix, (unsigned)yld * 4u) + ycol_off : TBN_OOB;
            ysoff = 0u;
          }
        } else if (!tile_full) {
          const int m = mrow0 + i * 32 + dm;
          voff = (m < p.M) ? vbase : TBN_OOB;  // the scalar offset is not bounds-checked: mask the row here
          if (RED) yvoff = (m < p.M) ? ybase : TBN_OOB;
        }
        if (EPI == 0 || EPI == 3) {
          if (p.flags & CONV_FLAG_ACCUM) v += tbn_llvm_buffer_load_f32(o_rsrc, (int)voff, (int)soff, 0);
          if (p.flags & CONV_FLAG_RELU) v = fmaxf(v, 0.f);
        }
        tbn_llvm_buffer_store_f32(v, o_rsrc, (int)voff, (int)soff, 0);
        if (RED) {
          // rows >= M / masked lanes: v may hold junk only where the store was masked too -> mask g the same way
          const float yv = tbn_llvm_buffer_load_f32(y_rsrc, (int)yvoff, (int)ysoff, 0);
          const float g = (yvoff != TBN_OOB && fmaf(yv, b_sc, b_sh) > 0.f) ? v : 0.f;
          s1 += g;
          s2 = fmaf(g, (yv - b_mu) * b_rs, s2);
        }
      }
    }
    }
    if (SUMS) {
      s1 += __shfl_xor(s1, 32);
      s2 += __shfl_xor(s2, 32);
      if (WM == 1) {   // the wave holds the whole tile: its column sums ARE the tile's partial row
        if (lhalf == 0 && col_ok) {
          if (EPI == 1) {
            p.stat_partial[((size_t)tm * 2 + 0) * p.Cout + col] = s1;
            p.stat_partial[((size_t)tm * 2 + 1) * p.Cout + col] = s2;
          } else if (red_on && lc < p.red[rs].C) {
            float* part = p.red[rs].partial + (size_t)(p.red_row0 + tm) * 2 * p.red[rs].C;
            part[lc] = s1;
            part[p.red[rs].C + lc] = s2;
          }
        }
      } else if (lhalf == 0) {
        red[(0 * 4 + wave) * BN + j * 32 + lrow] = s1;
        red[(1 * 4 + wave) * BN + j * 32 + lrow] = s2;
      }
    }
  }
  if (SUMS && WM != 1) {
    __syncthreads();
    if (tid < BN && n0 + tid < p.Cout) {
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        t1 += red[(0 * 4 + w) * BN + tid];
        t2 += red[(1 * 4 + w) * BN + tid];
      }
      if (EPI == 1) {
        p.stat_partial[((size_t)tm * 2 + 0) * p.Cout + n0 + tid] = t1;
        p.stat_partial[((size_t)tm * 2 + 1) * p.Cout + n0 + tid] = t2;
      } else {
        const int col = n0 + tid;
        int rs = 0;
        if (p.nred > 1 && col >= p.red[1].col_begin) rs = 1;
        if (p.nred > 2 && col >= p.red[2].col_begin) rs = 2;
        if (p.nred > 3 && col >= p.red[3].col_begin) rs = 3;
        const int lc = col - p.red[rs].col_begin;
        if (p.red[rs].y != nullptr && lc < p.red[rs].C) {
          float* part = p.red[rs].partial + (size_t)(p.red_row0 + tm) * 2 * p.red[rs].C;
          part[lc] = t1;
          part[p.red[rs].C + lc] = t2;
        }
      }
    }
  }
}

// EPI: 0 plain (+bias, optional ReLU / accumulate), 1 training-BN statistics, 2 eval-BN fold + ReLU,
//      3 plain with output scatter (parity phase of a strided data gradient)
// RED (EPI 0 / 3 only): the epilogue also forms the BN-backward reduce partials of the layers that produced the
//      activation whose gradient this launch finishes (RedSeg in tbn_kernels.h)
template <int MT, int NT, bool ROWMODE, int EPI, int STAGES, bool RED = false>
__device__ __forceinline__ void conv_igemm_body(const ConvP& p, const int bid, float* lds) {
  constexpr int BM = 128 * MT, BN = 32 * NT;
  constexpr int AR = 4 * MT;  // A rows per thread
  constexpr int TILE_F = (BM + BN) * LDT;  // floats per LDS stage (A rows then B rows)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // XCD-aware bijective remap: blocks b, b+8, ... share an XCD -> give them consecutive tiles
  const int nb = p.tiles_m * p.tiles_n;
  const int q8 = nb >> 3, r8 = nb & 7, xcd = bid & 7, idx = bid >> 3;
  const int nid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
  const int tm = nid / p.tiles_n, tn = nid - tm * p.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;

  // descriptors are built from kernel arguments only (wave-uniform): loads need no branches, an
  // out-of-image tap / out-of-range row is an out-of-range OFFSET and reads zeros
  const i32x4 in_rsrc = make_rsrc(p.in, p.in_bytes);
  const i32x4 wt_rsrc = make_rsrc(p.wt, p.wt_bytes);

  const int c4 = tid & 7, r0 = tid >> 3;
  unsigned a_off[AR];   // byte offset of the row's (n, iy0, ix0) pixel (ROWMODE: out of range for rows >= M)
  unsigned a_mask[AR];  // bit t: tap t of this row lies inside the image (unused in ROWMODE: the border is physical)
  // Row setup (VALU diet, round 3).  Eight threads (c4 = 0..7) share each tile row: decoding the rows per thread ran the
  // two magic divisions and the tap masks 8x per row, ~35 VALU x 4*MT rows per thread.  Now a pointwise launch (1x1,
  // stride 1, no padding: every fused 1x1 group) needs no decode at all -- the input pixel IS the output row -- and
  // every other launch decodes each row ONCE (thread t < BM takes row t) into an LDS table the threads then read.
  const bool pointwise = !ROWMODE && p.ntaps == 1 && p.in_sy == 1 && p.in_sx == 1 && p.ty0 == 0 && p.tx0 == 0 &&
                         p.OHs == p.H && p.OWs == p.W;
  if (pointwise) {
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      const int m = m0 + r0 + 32 * i;
      a_off[i] = (unsigned)m * (unsigned)p.in_ld * 4u;
      a_mask[i] = m < p.M ? 1u : 0u;
    }
  } else {
    unsigned* rowtab = reinterpret_cast<unsigned*>(lds);   // [BM][2]: the tiles are not staged yet
    if (tid < BM) {
      const int m = m0 + tid;
      unsigned mask = 0;
      unsigned off = ROWMODE ? TBN_OOB : 0u;
      if (m < p.M) {
        const uint32_t n = fdiv((uint32_t)m, p.div_ohw);
        const uint32_t rem = (uint32_t)m - n * p.div_ohw.d;
        const uint32_t a = fdiv(rem, p.div_ow);
        const uint32_t b = rem - a * p.div_ow.d;
        const int iy0 = (int)a * p.in_sy;
        const int ix0 = (int)b * p.in_sx;
        if (ROWMODE) {
          off = (unsigned)((((int)n * p.H + iy0) * p.W + ix0) * p.cp * 4);
        } else {
          // taps form a (tny x tnx) grid starting at (ty0, tx0): validity = row bits x column bits
          unsigned yb = 0, xb = 0;
#pragma unroll
          for (int r = 0; r < 3; ++r)
            if (r < p.tny && (unsigned)(iy0 + p.ty0 + r) < (unsigned)p.H) yb |= 1u << r;
          off = (unsigned)((((int)n * p.H + iy0) * p.W + ix0) * p.in_ld * 4);
#pragma unroll
          for (int c = 0; c < 3; ++c)
            if (c < p.tnx && (unsigned)(ix0 + p.tx0 + c) < (unsigned)p.W) xb |= 1u << c;
#pragma unroll
          for (int r = 0; r < 3; ++r)
            if ((yb >> r) & 1u) mask |= xb << (r * p.tnx);
        }
      }
      rowtab[2 * tid] = off;
      rowtab[2 * tid + 1] = mask;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      const uint2 e = *reinterpret_cast<const uint2*>(&rowtab[2 * (r0 + 32 * i)]);
      a_off[i] = e.x;
      a_mask[i] = e.y;
    }
    __syncthreads();   // everyone has read the table before the first tile store overwrites it
  }

  if (ABL(128)) {  // ablation: exit after the per-row setup
    if (a_off[0] == 0x12345u && a_mask[AR - 1] == 77u) p.seg[0].ptr[0] = 1.f;
    return;
  }
  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int ksteps = p.K >> 5;
  float4 ra[AR], rb[NT];
  const unsigned b_row = (unsigned)(n0 + r0) * (unsigned)p.Krow * 4u + (unsigned)c4 * 16u;

  // (tap, c0) of the NEXT tile to fetch advance incrementally: no division, tap tables are read with
  // scalar loads only when the tap changes
  int l_tap = 0, l_c0 = 0;
  unsigned l_toff = (unsigned)p.tap_off[0], l_koff = (unsigned)p.tap_koff[0] * 4u;
  unsigned a_cur[AR];  // per-row byte offset for the CURRENT tap (or OOB): recomputed only when the tap changes;
                       // the channel-chunk offset rides in the instruction's scalar offset -> no VALU per load
  unsigned b_voff[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) b_voff[i] = b_row + (unsigned)(32 * i) * (unsigned)p.Krow * 4u;
  const unsigned rm_pitch = (unsigned)(p.W * p.cp) * 4u;   // ROWMODE: bytes between the runs of a pixel (one image row)
  auto load_tiles = [&]() {
    if (!ROWMODE) {
      if (l_c0 == 0) {
        const unsigned toff = l_toff + (unsigned)c4 * 16u;
#pragma unroll
        for (int i = 0; i < AR; ++i) a_cur[i] = ((a_mask[i] >> l_tap) & 1u) ? a_off[i] + toff : TBN_OOB;
      }
      const unsigned soff = (unsigned)l_c0 * 4u;
#pragma unroll
      for (int i = 0; i < AR; ++i) ra[i] = buf_load4(in_rsrc, a_cur[i], soff);
    } else {
      // l_c0 counts the packed K floats: this lane's float4 sits in run t at float4 w4 (one division per K-step,
      // shared by the thread's AR rows; no masks -- the zero border is in the image)
      const uint32_t f = ((uint32_t)l_c0 >> 2) + (uint32_t)c4;
      const uint32_t t = fdiv(f, p.div_rl4);
      const unsigned roff = t * rm_pitch + (f - t * p.div_rl4.d) * 16u;
#pragma unroll
      for (int i = 0; i < AR; ++i) ra[i] = buf_load4(in_rsrc, a_off[i] + roff);   // rows >= M: 2^31 + roff stays out of range
    }
    const unsigned koff = l_koff + (unsigned)l_c0 * 4u;
#pragma unroll
    for (int i = 0; i < NT; ++i)  // rows >= Cout are beyond wt_bytes -> zeros
      rb[i] = buf_load4(wt_rsrc, b_voff[i], koff);
    l_c0 += 32;
    if (!ROWMODE && l_c0 == p.Cin) {
      l_c0 = 0;
      ++l_tap;
      if (l_tap < p.ntaps) {
        l_toff = (unsigned)p.tap_off[l_tap];
        l_koff = (unsigned)p.tap_koff[l_tap] * 4u;
      }
    }
  };
  auto store_tiles = [&](float* stage) {
    float* As = stage;
    float* Bs = stage + BM * LDT;
#pragma unroll
    for (int i = 0; i < AR; ++i) *reinterpret_cast<float4*>(&As[(r0 + 32 * i) * LDT + c4 * 4]) = ra[i];
#pragma unroll
    for (int i = 0; i < NT; ++i) *reinterpret_cast<float4*>(&Bs[(r0 + 32 * i) * LDT + c4 * 4]) = rb[i];
  };

  load_tiles();
  store_tiles(lds);
  __syncthreads();
  if (ABL(256)) {  // ablation: exit after the prologue tile
    if (lds[tid] == 12345.678f) p.seg[0].ptr[0] = 1.f;
    return;
  }

  const int lrow = lane & 31, lhalf = lane >> 5;
  // two LDS stages, ONE barrier per K-step: the stage written in step ks was last read in step ks-1,
  // and every wave has passed the barrier that ended step ks-1 before any wave writes it.
  // The fragment reads and MFMAs of a K-step are ONE basic block (no run-time conditions inside), so the
  // scheduler issues the ds_reads ahead of the MFMAs that hide them.  A ragged last N tile multiplies its
  // zero-filled weight rows (rows >= Cout are out-of-range loads) instead of branching around MFMAs.
  for (int ks = 0; ks < ksteps; ++ks) {
    const bool more = (ks + 1 < ksteps) && !ABL(4);   // flag 4: ablation, no loads in the loop
    if (more) load_tiles();  // global loads stay in flight under the MFMA phase
    const float* As = lds + (STAGES == 2 ? (ks & 1) * TILE_F : 0);
    const float* Bs = As + BM * LDT;
    float4 fa[2][MT], fb[2][NT];
    auto frag_load = [&](int buf, int kg) {
#pragma unroll
      for (int i = 0; i < MT; ++i)
        fa[buf][i] =
            *reinterpret_cast<const float4*>(&As[(wave * 32 * MT + i * 32 + lrow) * LDT + kg * 8 + lhalf * 4]);
#pragma unroll
      for (int j = 0; j < NT; ++j)
        fb[buf][j] = *reinterpret_cast<const float4*>(&Bs[(j * 32 + lrow) * LDT + kg * 8 + lhalf * 4]);
    };
    auto mfma_group = [&](int buf) {
      // the 4 MFMAs of one accumulator stay back to back: a dependent MFMA is only free when it directly
      // follows its producer (scripts/ubench: alternating two accumulators halves the rate)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[buf][i].x, fb[buf][j].x, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[buf][i].y, fb[buf][j].y, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[buf][i].z, fb[buf][j].z, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[buf][i].w, fb[buf][j].w, acc[i][j], 0, 0, 0);
        }
    };
    // software pipeline over the four 8-wide k groups: the fragments of group g+1 are read while group g
    // multiplies; the sched_group_barrier chain pins that order (DS-read group, MFMA group, ...)
    if (!ABL(32)) {   // flag 32: ablation, no LDS fragment reads
      frag_load(0, 0);
      frag_load(1, 1);
    }
    if (!ABL(8)) mfma_group(0);   // flag 8: ablation, no MFMA
    if (!ABL(32)) frag_load(0, 2);
    if (!ABL(8)) mfma_group(1);
    if (!ABL(32)) frag_load(1, 3);
    if (!ABL(8)) mfma_group(0);
    if (!ABL(8)) mfma_group(1);
    if (!TBN_ABLATE) {
      constexpr int NR = MT + NT, NM = 4 * MT * NT;
      __builtin_amdgcn_sched_group_barrier(0x100, 2 * NR, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, NM, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, NR, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, NM, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, NR, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 2 * NM, 0);
      __builtin_amdgcn_sched_barrier(0);  // keep the barrier (and its lgkmcnt(0)) below the last MFMA group
    }
    if (STAGES == 2) {
      if (more) store_tiles(lds + ((ks + 1) & 1) * TILE_F);
      if (!ABL(64)) __syncthreads();   // flag 64: ablation, no barrier
    } else {  // single stage: everyone must be done reading before the tile is overwritten
      __syncthreads();
      if (more) {
        store_tiles(lds);
        __syncthreads();
      }
    }
  }

  conv_epilogue<MT, NT, EPI, RED>(p, acc, lds, tm, m0, n0);
}

template <int MT, int NT, bool ROWMODE, int EPI, int STAGES, bool RED = false>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvP p, RiderP rider) {
  __shared__ __attribute__((aligned(16))) float lds[STAGES * (128 * MT + 32 * NT) * LDT];
  TBN_RIDER_DISPATCH(rider, bid)
  conv_igemm_body<MT, NT, ROWMODE, EPI, STAGES, RED>(p, bid, lds);
}

// LDS-DMA primitive (see the LDS-DMA variant of the generic kernel further down for the protocol): lane l's 16 bytes land at
// (wave-uniform LDS address in M0) + 16 l
__device__ __forceinline__ void lds_dma16(i32x4 rsrc, unsigned lds_dst, unsigned voff, unsigned soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
               :
               : "s"(lds_dst), "v"(voff), "s"(rsrc), "s"(soff)
               : "memory");   // (m0 cannot be named as a clobber: clang rejects it as a reserved register; nothing else in this
                              //  kernel keeps a value in M0 -- no readlane / movrel / LDS-direct uses)
}

// Round 6: the LDS-halo kernel's WEIGHT tiles (one 32*NT x 32 tile per tap) go global -> LDS by LDS-DMA for the MT = 1 tiles -- no
// VGPR round trip, no ds_write, the store's issue slots go to the MFMAs (scripts/ubench/kstep_cost.hip: the tile store costs the
// K-step 3.5 % of the peak).  Measured (profiles/r06_ab_halo_dma_b.txt, r06_conv_variants_halo_dma_b.txt): steady state <1,1> +0.8 %,
// <1,2> +2.6 %, <1,3> +0.7 %; <2,x> -0.4 ... -1.2 % (they stay register-staged); one-stream conv stage of the bench +1.1 points
// (3 of 3), three-stream step +-0.  -DTBN_HALO_DMA_B=0 restores the register-staged form for every tile.
#ifndef TBN_HALO_DMA_B
#define TBN_HALO_DMA_B 1
#endif

// ------------------------------------------------------------------------------------------
// 3x3 / stride 1 / pad 1 convolution (forward of such a layer, and its data gradient, which has the same form) with
// the INPUT PATCH STAGED ONCE PER 32-CHANNEL CHUNK.  The generic body above gathers an im2col A tile per filter tap:
// nine global -> LDS copies of (almost) the same pixels.  Here the flat NHWC pixel range of a 128*MT-row output tile,
// widened by W + 1 pixels on both sides (a "flat halo": BM + 2W + 2 consecutive pixels, contiguous in HBM), is loaded
// once per channel chunk and the nine taps read it at nine row shifts:
//   tap (dy, dx) of output pixel m  ->  halo row (m - m0) + (W + 1) + dy*W + dx.
// Where that shift leaves the image (top / bottom row, left / right column, frame boundary) the lane reads a row of
// zeros instead: the nine per-lane row addresses are loop invariants computed once.  Global -> LDS traffic of the A
// operand drops by 9*BM / (BM + 2W + 2) (4.8x at W = 56, 6.2x at W = 28 for BM = 128), its load / ds_write instructions
// and the per-tap address VALU likewise.  Weights stream as before: one (32*NT) x 32 B tile per tap, double buffered.
// LDS: [BM + 2W + 3 rows] A (one buffer: the next chunk waits in registers) + 2 x [32*NT rows] B, pitch LDT.
template <int MT, int NT, int EPI, bool RED>
__device__ __forceinline__ void conv_halo_body(const ConvP& p, const int bid, float* lds) {
  constexpr int BM = 128 * MT, BN = 32 * NT;
  constexpr int NJ = (BM + 2 * 64 + 2 + 31) / 32;   // float4 slots per thread for a halo of up to W = 64
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nb = p.tiles_m * p.tiles_n;
  const int q8 = nb >> 3, r8 = nb & 7, xcd = bid & 7, idx = bid >> 3;
  const int nid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
  const int tm = nid / p.tiles_n, tn = nid - tm * p.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int W = p.W, HR = BM + 2 * W + 2;   // halo rows; row HR is the zero row
  float* As = lds;
  float* Bs0 = lds + (HR + 1) * LDT;

  const i32x4 in_rsrc = make_rsrc(p.in, p.in_bytes);
  const i32x4 wt_rsrc = make_rsrc(p.wt, p.wt_bytes);
  const int c4 = tid & 7, r0 = tid >> 3;
  const int lrow = lane & 31, lhalf = lane >> 5;

  // per-lane LDS byte address of each tap's A fragment row (or the zero row)
  unsigned fa_off[MT][9];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int lr = wave * 32 * MT + i * 32 + lrow;
    const int m = m0 + lr;
    const uint32_t n = fdiv((uint32_t)m, p.div_ohw);
    const uint32_t rem = (uint32_t)m - n * p.div_ohw.d;
    const int y = (int)fdiv(rem, p.div_ow);
    const int x = (int)rem - y * W;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int dy = t / 3 - 1, dx = t % 3 - 1;
      const bool ok = (m < p.M) && ((unsigned)(y + dy) < (unsigned)p.H) && ((unsigned)(x + dx) < (unsigned)W);
      const int row = ok ? lr + (W + 1) + dy * W + dx : HR;
      fa_off[i][t] = (unsigned)(row * LDT + lhalf * 4) * 4u;
    }
  }
  // halo slot j of this thread: row (tid >> 3) + 32 j, 16-B column c4
  const int pix0 = m0 - (W + 1) + r0;
  float4 ha[NJ];
  auto load_halo = [&](int c0) {
    const unsigned soff = (unsigned)c0 * 4u;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int h = r0 + 32 * j, pix = pix0 + 32 * j;
      const bool ok = (h < HR) && ((unsigned)pix < (unsigned)p.M);
      ha[j] = buf_load4(in_rsrc, ok ? (unsigned)pix * (unsigned)p.in_ld * 4u + (unsigned)c4 * 16u : TBN_OOB, soff);
    }
  };
  auto store_halo = [&](int c0) {
#if TBN_DIAG
    if (p.fold_scale != nullptr) {   // cost probe: BN apply + ReLU of the producer layer at staging time (zero padding kept)
      const float4 fs = *reinterpret_cast<const float4*>(p.fold_scale + c0 + c4 * 4);
      const float4 fh = *reinterpret_cast<const float4*>(p.fold_shift + c0 + c4 * 4);
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const bool ok = (r0 + 32 * j < HR) && ((unsigned)(pix0 + 32 * j) < (unsigned)p.M);
        ha[j].x = ok ? fmaxf(fmaf(ha[j].x, fs.x, fh.x), 0.f) : 0.f;
        ha[j].y = ok ? fmaxf(fmaf(ha[j].y, fs.y, fh.y), 0.f) : 0.f;
        ha[j].z = ok ? fmaxf(fmaf(ha[j].z, fs.z, fh.z), 0.f) : 0.f;
        ha[j].w = ok ? fmaxf(fmaf(ha[j].w, fs.w, fh.w), 0.f) : 0.f;
      }
    }
#endif
    (void)c0;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
      if (r0 + 32 * j < HR) *reinterpret_cast<float4*>(&As[(r0 + 32 * j) * LDT + c4 * 4]) = ha[j];
  };
  constexpr bool DMAB = (TBN_HALO_DMA_B != 0) && MT == 1;
  float4 rb[NT];
  unsigned b_voff[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
    b_voff[i] = (unsigned)(n0 + r0 + 32 * i) * (unsigned)p.Krow * 4u + (unsigned)c4 * 16u;   // rows >= Cout: beyond wt_bytes
  auto load_b = [&](int t, int c0) {
    const unsigned koff = (unsigned)(t * p.Cin + c0) * 4u;
#pragma unroll
    for (int i = 0; i < NT; ++i) rb[i] = buf_load4(wt_rsrc, b_voff[i], koff);
  };
  auto store_b = [&](float* Bs) {
#pragma unroll
    for (int i = 0; i < NT; ++i) *reinterpret_cast<float4*>(&Bs[(r0 + 32 * i) * LDT + c4 * 4]) = rb[i];
  };
  // DMAB: the weight tile of a tap goes global -> LDS directly (no VGPR round trip, no ds_write): rows are 128 B, unpadded,
  // 16-B slot q of row r at slot q ^ ((r >> 1) & 7) (the swizzle of conv_dma_body, applied on the source address and again
  // on the fragment read); a wave instruction writes 8 rows
  const int dr = lane >> 3, dslot = lane & 7;
  unsigned bd_voff[NT], fb_addr[NT][4];
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    const int r = (wave + 4 * i) * 8 + dr;
    bd_voff[i] = (unsigned)(n0 + r) * (unsigned)p.Krow * 4u + (unsigned)((dslot ^ ((r >> 1) & 7)) << 4);
  }
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int r = j * 32 + lrow;
#pragma unroll
    for (int kg = 0; kg < 4; ++kg) fb_addr[j][kg] = (unsigned)(r * 128 + (((2 * kg + lhalf) ^ ((r >> 1) & 7)) << 4));
  }
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  const unsigned ldsB = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)reinterpret_cast<char*>(Bs0));
  auto dma_b = [&](int t, int c0, int stage) {
    const unsigned koff = (unsigned)__builtin_amdgcn_readfirstlane((t * p.Cin + c0) * 4);
#pragma unroll
    for (int i = 0; i < NT; ++i) lds_dma16(wt_rsrc, ldsB + (unsigned)(stage * BN * 128 + (wv + 4 * i) * 1024), bd_voff[i], koff);
  };

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // prologue: zero row, halo of chunk 0, B tile of (tap 0, chunk 0)
  load_halo(0);
  if (DMAB)
    dma_b(0, 0, 0);
  else
    load_b(0, 0);
  if (tid < LDT) As[HR * LDT + tid] = 0.f;
  store_halo(0);
  if (DMAB)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else
    store_b(Bs0);
  __syncthreads();

  const int nchunks = p.Cin >> 5;
  const char* As_b = reinterpret_cast<const char*>(As);
  int ks = 0;
  // One 32-channel chunk = nine taps.  NEXT (a type, so every `if` on it is resolved at compile time): there is another
  // chunk behind this one -- its halo is fetched at the top and the B tile of its tap 0 during tap 8.  The last chunk is a
  // second copy of the body WITHOUT those parts rather than the same code under `if (c + 1 < nchunks)`: with the prefetches
  // under a run-time condition the compiler's wait-count pass must assume that the B-tile loads of the previous trip may
  // still be in flight where the first ds_read of a trip reuses their registers, and put `s_waitcnt vmcnt(1)` there --
  // right behind the thirteen halo loads just issued, i.e. every chunk started by waiting for its successor's halo.
  auto chunk = [&](const int c, auto next_tag) {
    constexpr bool NEXT = decltype(next_tag)::value;
    if (NEXT) load_halo((c + 1) * 32);   // lands during tap 0 (the first B-tile wait is behind it in vmcnt order)
#pragma unroll
    for (int t = 0; t < 9; ++t, ++ks) {
      const bool more = NEXT || t < 8;
      if (more) {
        // DMAB: straight into the other stage (last read in the previous tap: every wave is past that tap's barrier)
        if (DMAB)
          dma_b(t < 8 ? t + 1 : 0, t < 8 ? c * 32 : (c + 1) * 32, (ks + 1) & 1);
        else
          load_b(t < 8 ? t + 1 : 0, t < 8 ? c * 32 : (c + 1) * 32);
      }
      const float* Bs = Bs0 + (ks & 1) * (BN * LDT);
      const char* Bd = reinterpret_cast<const char*>(Bs0) + (ks & 1) * (BN * 128);
      float4 fa[2][MT], fb[2][NT];
      auto frag_load = [&](int buf, int kg) {
#pragma unroll
        for (int i = 0; i < MT; ++i) fa[buf][i] = *reinterpret_cast<const float4*>(As_b + fa_off[i][t] + kg * 32);
#pragma unroll
        for (int j = 0; j < NT; ++j)
          fb[buf][j] = DMAB ? *reinterpret_cast<const float4*>(Bd + fb_addr[j][kg])
                            : *reinterpret_cast<const float4*>(&Bs[(j * 32 + lrow) * LDT + kg * 8 + lhalf * 4]);
      };
      auto mfma_group = [&](int buf) {
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int i = 0; i < MT; ++i) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[buf][i].x, fb[buf][j].x, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[buf][i].y, fb[buf][j].y, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[buf][i].z, fb[buf][j].z, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[buf][i].w, fb[buf][j].w, acc[i][j], 0, 0, 0);
          }
      };
      frag_load(0, 0);
      frag_load(1, 1);
      mfma_group(0);
      frag_load(0, 2);
      mfma_group(1);
      frag_load(1, 3);
      mfma_group(0);
      mfma_group(1);
      {
        constexpr int NR = MT + NT, NM = 4 * MT * NT;
        // (the B-tile loads of the next tap are left to the scheduler, which sinks them to 4-12 MFMAs in front of the
        // ds_write that waits for them: pinning them to the top of the tap with a VMEM group measured 0-2 % SLOWER on
        // warm clocks, 25 % on one <1,3> shape -- the weights are L2 hits and the early loads lengthen live ranges)
        __builtin_amdgcn_sched_group_barrier(0x100, 2 * NR, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, NM, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, NR, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, NM, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, NR, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 2 * NM, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (DMAB)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMA of the next tap's weights has landed (and so has
                                                           // the next chunk's halo prefetch issued at the top of tap 0)
      else if (more)
        store_b(Bs0 + ((ks + 1) & 1) * (BN * LDT));
      __syncthreads();
    }
    if (NEXT) {   // every wave is past its last read of this chunk's halo (barrier above)
      store_halo((c + 1) * 32);
      __syncthreads();
    }
  };
  for (int c = 0; c + 1 < nchunks; ++c) chunk(c, std::true_type{});
  chunk(nchunks - 1, std::false_type{});
  conv_epilogue<MT, NT, EPI, RED>(p, acc, lds, tm, m0, n0);
}

template <int MT, int NT, int EPI, bool RED>
__global__ __launch_bounds__(256) void conv_halo_kernel(ConvP p, RiderP rider) {
  extern __shared__ __attribute__((aligned(16))) float dyn_lds[];
  TBN_RIDER_DISPATCH(rider, bid)
  conv_halo_body<MT, NT, EPI, RED>(p, bid, dyn_lds);
}

// ------------------------------------------------------------------------------------------
// LDS-DMA variant of the generic implicit GEMM (1x1 / 3x3, any stride; not the stem): the A and B tiles go from
// global memory STRAIGHT into LDS (`buffer_load_dwordx4 ... lds`: no VGPR round trip, no ds_write issue, 4*MT + NT
// fewer VGPR quads), two stages, one barrier per K-step.  An LDS-DMA instruction writes lane l's 16 bytes at
// (wave-uniform base) + 16 l, i.e. 8 rows x 128 B per wave instruction: rows are UNPADDED and conflict-free b128
// fragment reads come from an XOR swizzle instead -- 16-B slot q of tile row r lives at slot q ^ ((r >> 1) & 7); the
// swizzle is applied on the global SOURCE address of each lane (lane (row l >> 3, slot l & 7) fetches chunk
// (l & 7) ^ f(row)) and again on the fragment read.  Out-of-image taps / rows beyond M or Cout are out-of-range buffer
// offsets: the DMA writes zeros.
// The DMA is issued through inline asm: hipcc would otherwise treat each LDS-DMA as a pending LDS write and put an
// `s_waitcnt vmcnt(0)` in front of the next fragment read -- draining the prefetch it is supposed to overlap.  The
// asm statement is invisible to that bookkeeping; the wait is placed by hand before the barrier that publishes the
// stage (cdna_hip_programming.md 5.7: M0 is written in the same statement that uses it).

template <int MT, int NT, int EPI, bool RED>
__device__ __forceinline__ void conv_dma_body(const ConvP& p, const int bid, float* lds) {
  constexpr int BM = 128 * MT, BN = 32 * NT;
  constexpr int AR = 4 * MT;                 // 8-row DMA instructions per wave for the A tile
  constexpr int TILE_B = (BM + BN) * 128;    // bytes per LDS stage (A rows then B rows, 128 B each)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nb = p.tiles_m * p.tiles_n;
  const int q8 = nb >> 3, r8 = nb & 7, xcd = bid & 7, idx = bid >> 3;
  const int nid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
  const int tm = nid / p.tiles_n, tn = nid - tm * p.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const i32x4 in_rsrc = make_rsrc(p.in, p.in_bytes);
  const i32x4 wt_rsrc = make_rsrc(p.wt, p.wt_bytes);

  const int rl = lane >> 3, slot = lane & 7;   // row within the 8-row group / 16-B slot this lane's bytes land in
  unsigned a_off[AR], a_mask[AR];
  {
    // each tile row is decoded once (thread t < BM takes row t) into an LDS table -- eight lanes share a row; decoding
    // per lane repeated the two magic divisions and the tap masks 8x (see conv_igemm_body)
    unsigned* rowtab = reinterpret_cast<unsigned*>(lds);   // [BM][2]: no DMA has been issued yet
    if (tid < BM) {
      const int m = m0 + tid;
      unsigned mask = 0;
      int off = 0;
      if (m < p.M) {
        const uint32_t n = fdiv((uint32_t)m, p.div_ohw);
        const uint32_t rem = (uint32_t)m - n * p.div_ohw.d;
        const uint32_t a = fdiv(rem, p.div_ow);
        const uint32_t b = rem - a * p.div_ow.d;
        const int iy0 = (int)a * p.in_sy, ix0 = (int)b * p.in_sx;
        unsigned yb = 0, xb = 0;
#pragma unroll
        for (int rr = 0; rr < 3; ++rr)
          if (rr < p.tny && (unsigned)(iy0 + p.ty0 + rr) < (unsigned)p.H) yb |= 1u << rr;
#pragma unroll
        for (int c = 0; c < 3; ++c)
          if (c < p.tnx && (unsigned)(ix0 + p.tx0 + c) < (unsigned)p.W) xb |= 1u << c;
#pragma unroll
        for (int rr = 0; rr < 3; ++rr)
          if ((yb >> rr) & 1u) mask |= xb << (rr * p.tnx);
        off = (((int)n * p.H + iy0) * p.W + ix0) * p.in_ld * 4;
      }
      rowtab[2 * tid] = (unsigned)off;
      rowtab[2 * tid + 1] = mask;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      const int r = (wave * AR + i) * 8 + rl;    // tile row
      const uint2 e = *reinterpret_cast<const uint2*>(&rowtab[2 * r]);
      a_off[i] = e.x + (unsigned)((slot ^ ((r >> 1) & 7)) << 4);   // swizzled source chunk (masked rows never load)
      a_mask[i] = e.y;
    }
    __syncthreads();   // the table is dead: the first DMA may overwrite it
  }
  unsigned b_voff[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    const int r = (wave + 4 * i) * 8 + rl;     // B tile row (= output channel n0 + r); rows >= Cout: beyond wt_bytes
    b_voff[i] = (unsigned)(n0 + r) * (unsigned)p.Krow * 4u + (unsigned)((slot ^ ((r >> 1) & 7)) << 4);
  }
  // wave-uniform LDS byte addresses of this wave's DMA instructions (stage 0)
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  char* const lds_b = reinterpret_cast<char*>(lds);
  // (the low 32 bits of a generic pointer into LDS are the LDS byte offset)
  const unsigned lds0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)lds_b);

  // unit-stride launches only (no parity phases): tap t's weights start at K offset t * Cin, so the weight-row byte
  // offset of a K-step is a plain scalar counter
  int l_tap = 0, l_c0 = 0;
  unsigned l_toff = (unsigned)p.tap_off[0], k_byte = 0u;
  unsigned a_cur[AR];
  auto issue_tiles = [&](int stage) {
    if (l_c0 == 0) {
#pragma unroll
      for (int i = 0; i < AR; ++i) a_cur[i] = ((a_mask[i] >> l_tap) & 1u) ? a_off[i] + l_toff : TBN_OOB;
    }
    // (the loop-carried counters end up in VGPRs -- SIFixSGPRCopies -- and an "s" asm operand is not legalised: readfirstlane)
    const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane(l_c0 * 4);
    const unsigned st = lds0 + (unsigned)(stage * TILE_B);
#pragma unroll
    for (int i = 0; i < AR; ++i) lds_dma16(in_rsrc, st + (unsigned)((wv * AR + i) * 1024), a_cur[i], soff);
    const unsigned koff = (unsigned)__builtin_amdgcn_readfirstlane((int)k_byte);
#pragma unroll
    for (int i = 0; i < NT; ++i) lds_dma16(wt_rsrc, st + (unsigned)(BM * 128 + (wv + 4 * i) * 1024), b_voff[i], koff);
    k_byte += 128u;
    l_c0 += 32;
    if (l_c0 == p.Cin) {
      l_c0 = 0;
      ++l_tap;
      if (l_tap < p.ntaps) l_toff = (unsigned)p.tap_off[l_tap];
    }
  };

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // fragment addresses (bytes, stage 0): logical 16-B chunk (2 kg + lhalf) of row r sits at slot chunk ^ f(r)
  const int lrow = lane & 31, lhalf = lane >> 5;
  unsigned fa_addr[MT][4], fb_addr[NT][4];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int r = wave * 32 * MT + i * 32 + lrow;
#pragma unroll
    for (int kg = 0; kg < 4; ++kg) fa_addr[i][kg] = (unsigned)(r * 128 + (((2 * kg + lhalf) ^ ((r >> 1) & 7)) << 4));
  }
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int r = j * 32 + lrow;
#pragma unroll
    for (int kg = 0; kg < 4; ++kg)
      fb_addr[j][kg] = (unsigned)(BM * 128 + r * 128 + (((2 * kg + lhalf) ^ ((r >> 1) & 7)) << 4));
  }

  const int ksteps = p.K >> 5;
  issue_tiles(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();   // publishes stage 0

  auto compute = [&](const char* sb) {
    float4 fa[2][MT], fb[2][NT];
    auto frag_load = [&](int buf, int kg) {
#pragma unroll
      for (int i = 0; i < MT; ++i) fa[buf][i] = *reinterpret_cast<const float4*>(sb + fa_addr[i][kg]);
#pragma unroll
      for (int j = 0; j < NT; ++j) fb[buf][j] = *reinterpret_cast<const float4*>(sb + fb_addr[j][kg]);
    };
    auto mfma_group = [&](int buf) {
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[buf][i].x, fb[buf][j].x, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[buf][i].y, fb[buf][j].y, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[buf][i].z, fb[buf][j].z, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[buf][i].w, fb[buf][j].w, acc[i][j], 0, 0, 0);
        }
    };
    frag_load(0, 0);
    frag_load(1, 1);
    mfma_group(0);
    frag_load(0, 2);
    mfma_group(1);
    frag_load(1, 3);
    mfma_group(0);
    mfma_group(1);
    {
      constexpr int NR = MT + NT, NM = 4 * MT * NT;
      __builtin_amdgcn_sched_group_barrier(0x100, 2 * NR, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, NM, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, NR, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, NM, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, NR, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 2 * NM, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // two K-steps per iteration: the stage is a compile-time constant (fragment offsets stay immediates)
  for (int ks = 0; ks < ksteps; ks += 2) {
    if (ks + 1 < ksteps) issue_tiles(1);     // stage 1 was last read in step ks - 1: every wave is past that barrier
    compute(lds_b);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMA of the next stage has landed ...
    __syncthreads();                                    // ... and so has everyone else's
    if (ks + 1 >= ksteps) break;
    if (ks + 2 < ksteps) issue_tiles(0);
    compute(lds_b + TILE_B);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  conv_epilogue<MT, NT, EPI, RED>(p, acc, lds, tm, m0, n0);
}

template <int MT, int NT, int EPI, bool RED>
__global__ __launch_bounds__(256) void conv_dma_kernel(ConvP p, RiderP rider) {
  __shared__ __attribute__((aligned(1024))) float lds[2 * (128 * MT + 32 * NT) * 32];
  TBN_RIDER_DISPATCH(rider, bid)
  conv_dma_body<MT, NT, EPI, RED>(p, bid, lds);
}

// ------------------------------------------------------------------------------------------
// Small-M GEMMs (the 7x7 / 8x8 maps: M = 4704 rows at 96 frames; the head Linear layers: M = 96): with 128-row tiles
// the grid is 37 M-tiles x a few N-tiles -- e.g. 259 workgroups for Cout = 224, one round of 256 CUs plus three
// stragglers that double the launch time.  Here a workgroup owns a (32*MT) x (32*NT) tile and its FOUR WAVES SPLIT K
// (wave w takes the 32-float K chunks w, w+4, ...): 4x more, 4x smaller tiles (1029 for that layer) that balance over the
// CUs, and K = 9 * Cin is long enough that a quarter of it still amortises the tile's prologue.  Each wave stages its own
// A / B chunk in a wave-private LDS tile -- no workgroup barrier in the K loop (wave-private LDS operations execute
// in program order) -- the four partial tiles are summed through LDS in fixed order (deterministic) and wave 0 runs the
// usual epilogue on the finished tile (conv_epilogue with WM = 1: partial-sum rows go straight to global memory).
template <int MT, int NT, int EPI, bool RED>
__device__ __forceinline__ void conv_sk4_body(const ConvP& p, const int bid, float* lds) {
  constexpr int BM = 32 * MT, BN = 32 * NT;
  constexpr int AR = 4 * MT, BR = 4 * NT;   // float4 loads per lane per K-step: 8 tile rows per wave instruction
  constexpr int TILE_F = (BM + BN) * LDT;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nb = p.tiles_m * p.tiles_n;
  const int q8 = nb >> 3, r8 = nb & 7, xcd = bid & 7, idx = bid >> 3;
  const int nid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
  const int tm = nid / p.tiles_n, tn = nid - tm * p.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const i32x4 in_rsrc = make_rsrc(p.in, p.in_bytes);
  const i32x4 wt_rsrc = make_rsrc(p.wt, p.wt_bytes);

  const int c4 = lane & 7, r0 = lane >> 3;
  unsigned a_off[AR], a_mask[AR];
  {
    // the four waves stage the SAME 32*MT rows (different K chunks): decoded once per row into an LDS table instead of
    // by every lane of every wave (32x redundant: ~35 VALU x 4*MT rows per lane)
    unsigned* rowtab = reinterpret_cast<unsigned*>(lds);   // [BM][2]: nothing is staged yet
    if (tid < BM) {
      const int m = m0 + tid;
      unsigned mask = 0, off = 0;
      if (m < p.M) {
        const uint32_t n = fdiv((uint32_t)m, p.div_ohw);
        const uint32_t rem = (uint32_t)m - n * p.div_ohw.d;
        const uint32_t a = fdiv(rem, p.div_ow);
        const uint32_t b = rem - a * p.div_ow.d;
        const int iy0 = (int)a * p.in_sy, ix0 = (int)b * p.in_sx;
        unsigned yb = 0, xb = 0;
#pragma unroll
        for (int r = 0; r < 3; ++r)
          if (r < p.tny && (unsigned)(iy0 + p.ty0 + r) < (unsigned)p.H) yb |= 1u << r;
        off = (unsigned)((((int)n * p.H + iy0) * p.W + ix0) * p.in_ld * 4);
#pragma unroll
        for (int c = 0; c < 3; ++c)
          if (c < p.tnx && (unsigned)(ix0 + p.tx0 + c) < (unsigned)p.W) xb |= 1u << c;
#pragma unroll
        for (int r = 0; r < 3; ++r)
          if ((yb >> r) & 1u) mask |= xb << (r * p.tnx);
      }
      rowtab[2 * tid] = off;
      rowtab[2 * tid + 1] = mask;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      const uint2 e = *reinterpret_cast<const uint2*>(&rowtab[2 * (r0 + 8 * i)]);
      a_off[i] = e.x;
      a_mask[i] = e.y;
    }
    __syncthreads();   // the table is dead: the waves' private tiles may overwrite it
  }
  unsigned b_voff[BR];
#pragma unroll
  for (int i = 0; i < BR; ++i) b_voff[i] = (unsigned)(n0 + r0 + 8 * i) * (unsigned)p.Krow * 4u + (unsigned)c4 * 16u;

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int ksteps = p.K >> 5, cpt = p.Cin >> 5;
  int l_tap = wave / cpt, l_c0 = (wave - l_tap * cpt) * 32;   // (tap, channel offset) of this wave's next chunk (scalar)
  float* As = lds + wave * TILE_F;
  float* Bs = As + BM * LDT;
  float4 ra[AR], rb[BR];
  auto load_tiles = [&]() {
    const unsigned toff = (unsigned)p.tap_off[l_tap] + (unsigned)c4 * 16u, soff = (unsigned)l_c0 * 4u;
#pragma unroll
    for (int i = 0; i < AR; ++i) ra[i] = buf_load4(in_rsrc, ((a_mask[i] >> l_tap) & 1u) ? a_off[i] + toff : TBN_OOB, soff);
    const unsigned koff = ((unsigned)p.tap_koff[l_tap] + (unsigned)l_c0) * 4u;
#pragma unroll
    for (int i = 0; i < BR; ++i) rb[i] = buf_load4(wt_rsrc, b_voff[i], koff);   // rows >= Cout: beyond wt_bytes -> zeros
    l_c0 += 128;                 // four chunks on
    while (l_c0 >= p.Cin) {
      l_c0 -= p.Cin;
      ++l_tap;
    }
  };
  auto store_tiles = [&]() {
#pragma unroll
    for (int i = 0; i < AR; ++i) *reinterpret_cast<float4*>(&As[(r0 + 8 * i) * LDT + c4 * 4]) = ra[i];
#pragma unroll
    for (int i = 0; i < BR; ++i) *reinterpret_cast<float4*>(&Bs[(r0 + 8 * i) * LDT + c4 * 4]) = rb[i];
  };
  const int lrow = lane & 31, lhalf = lane >> 5;
  if (wave < ksteps) {
    load_tiles();
    store_tiles();
  }
  for (int q = wave; q < ksteps; q += 4) {
    const bool more = q + 4 < ksteps;
    __builtin_amdgcn_wave_barrier();
    if (more) load_tiles();   // in flight under the MFMA phase
    float4 fa[2][MT], fb[2][NT];
    auto frag_load = [&](int buf, int kg) {
#pragma unroll
      for (int i = 0; i < MT; ++i)
        fa[buf][i] = *reinterpret_cast<const float4*>(&As[(i * 32 + lrow) * LDT + kg * 8 + lhalf * 4]);
#pragma unroll
      for (int j = 0; j < NT; ++j)
        fb[buf][j] = *reinterpret_cast<const float4*>(&Bs[(j * 32 + lrow) * LDT + kg * 8 + lhalf * 4]);
    };
    auto mfma_group = [&](int buf) {
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[buf][i].x, fb[buf][j].x, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[buf][i].y, fb[buf][j].y, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[buf][i].z, fb[buf][j].z, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[buf][i].w, fb[buf][j].w, acc[i][j], 0, 0, 0);
        }
    };
    frag_load(0, 0);
    frag_load(1, 1);
    mfma_group(0);
    frag_load(0, 2);
    mfma_group(1);
    frag_load(1, 3);
    mfma_group(0);
    mfma_group(1);
    {
      constexpr int NR = MT + NT, NM = 4 * MT * NT;
      __builtin_amdgcn_sched_group_barrier(0x100, 2 * NR, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, NM, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, NR, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, NM, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, NR, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 2 * NM, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_wave_barrier();
    if (more) store_tiles();   // behind this step's fragment reads in the wave's LDS queue
  }
  __syncthreads();   // the reduction buffer overlays every wave's tile
  float* red = lds;  // [3 waves][MT*NT][16][64 lanes]
  if (wave > 0) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) red[(((wave - 1) * MT * NT + i * NT + j) * 16 + e) * 64 + lane] = acc[i][j][e];
  }
  __syncthreads();
  if (wave > 0) return;
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int o = ((i * NT + j) * 16 + e) * 64 + lane;
        acc[i][j][e] = ((acc[i][j][e] + red[o]) + red[MT * NT * 1024 + o]) + red[2 * MT * NT * 1024 + o];
      }
  conv_epilogue<MT, NT, EPI, RED, 1>(p, acc, lds, tm, m0, n0);
}

template <int MT, int NT, int EPI, bool RED>
__global__ __launch_bounds__(256) void conv_sk4_kernel(ConvP p, RiderP rider) {
  __shared__ __attribute__((aligned(16))) float lds[4 * (32 * MT + 32 * NT) * LDT];
  TBN_RIDER_DISPATCH(rider, bid)
  conv_sk4_body<MT, NT, EPI, RED>(p, bid, lds);
}

// The four output-parity phases of a stride-2 data gradient in ONE launch: each phase alone is a small GEMM
// (M / 4 rows, 1-4 taps) that leaves most CUs idle; a workgroup finds its phase by a scalar scan.
struct ConvPhases {
  ConvP ph[4];
  int blk0[5];
  int n;
};
template <int MT, int NT, int STAGES, bool RED = false>
__global__ __launch_bounds__(256) void conv_igemm_phases_kernel(ConvPhases q, RiderP rider) {
  __shared__ __attribute__((aligned(16))) float lds[STAGES * (128 * MT + 32 * NT) * LDT];
  TBN_RIDER_DISPATCH(rider, bid)
  int ph = 0;
  while (ph + 1 < q.n && bid >= q.blk0[ph + 1]) ++ph;
  conv_igemm_body<MT, NT, false, 3, STAGES, RED>(q.ph[ph], bid - q.blk0[ph], lds);
}

// ------------------------------------------------------------------------------------------
// Weight gradient:  dW[co][tap][ci] = sum_m dy[m][co] * x[pix(m) + tap][ci]
// GEMM with the *pixel* index as the reduction dim.  Both operands are read as K-major rows
// (pixel rows, channel-contiguous) and used straight from a [k][32*T+4] LDS tile with
// ds_read_b32 fragments (lane -> (k = lane>>5, i = lane&31) is conflict-free).  Each of the 4
// waves of a workgroup reduces a different quarter of the workgroup's pixel range over the same
// (32*MT co) x (32*NT ci) tile of one filter tap; the 4 partial tiles are summed through LDS
// in fixed order (deterministic), then stored to `out` (final dW or a split-K slab).
// MODE 0: any filter / stride / pad (the x row of an output pixel is found from its in-frame index, see below);
//      1: the packed-row stem (ROWMODE of conv_igemm_body: the "Cin" columns are R runs of `rl` floats of a physically
//         zero-padded image, no masks);  2: pointwise (1x1 / stride 1 / pad 0, and the Linear layers): x row = dy row.
// VALU diet (the fp32 MFMA shares its issue port with the VALU: every VALU instruction in this loop is ~4 of an MFMA's 64
// cycles, and a (64-row) step of a 64 x 64 tile is only 32 MFMAs).  A lane owns ONE pixel row of the 16-row step
// (row = l >> 2) and the float4 columns {q, q+4, q+8, ...} (q = l & 3).  Round 2 decoded the row from scratch every
// step (two 64-bit magic divisions, bounds compares, one select per load for ragged column tiles: 38-46 VALU per
// step, 1.3 per MFMA for the 64 x 64 tile, 2.4 for 64 x 32).  Now the row state advances incrementally:
//   * dy: byte offset += 64 rows; rows >= M lie beyond the buffer extent (the hardware returns zeros), a split never
//     ends inside a step (rows_per_split is a multiple of 64), column offsets ride in the instruction's immediate;
//     ragged column tiles read finite neighbours / zeros beyond the extent into accumulator rows that are never stored;
//   * x: the in-frame output pixel index pp (+= 64 mod OH*OW with one conditional wrap, the frame byte base follows)
//     gives (oy, ox) with ONE v_mul_hi (exact: pp * OW < 2^32), the input pixel with 24-bit multiply-adds;
//   pointwise layers need none of it (2 VALU per step), the general form 15, the stem 11 + one add per load.
template <int MT, int NT, int MODE>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradP p) {
  constexpr bool ROWMODE = (MODE == 1);
  constexpr int WA = 32 * MT + 4, WB = 32 * NT + 4;  // LDS pitches (a +16 pad removes the write conflicts but gains nothing)
  constexpr int KR = 16;                             // pixel rows per wave step
  constexpr int TILE = KR * (WA + WB);
  constexpr int LDSF = (4 * TILE > 4 * 1024 ? 4 * TILE : 4 * 1024);
  __shared__ __attribute__((aligned(16))) float lds[LDSF];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* At = lds + wave * TILE;
  float* Bt = At + KR * WA;

  // block -> (split, tap, tile_ci, tile_co), split slowest.  All tiles x taps of one split re-read the same
  // dy / x row slab, so the XCD-aware bijective remap (blocks b, b+8, ... share an XCD and its L2) hands each
  // XCD a CONTIGUOUS run of logical ids: a slab is then fetched into one L2 instead of all eight.
  int b;
  {
    const int nb = gridDim.x, bid = blockIdx.x;
    const int q8 = nb >> 3, r8 = nb & 7, xcd = bid & 7, idx = bid >> 3;
    b = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
  }
  const int tco = b % p.tiles_co;
  b /= p.tiles_co;
  const int tci = b % p.tiles_ci;
  b /= p.tiles_ci;
  const int tap = b % p.taps;
  const int split = b / p.taps;
  const int co0 = tco * 32 * MT, ci0 = tci * 32 * NT;
  int r, s;
  if (ROWMODE) {
    r = tap;
    s = 0;
  } else {
    r = tap / p.S;
    s = tap - r * p.S;
  }
  const int pbeg = split * p.rows_per_split;

  // K-split accumulators per sub-tile so that >= 4 independent accumulators rotate (see the main loop)
  constexpr int KS = (MT * NT >= 4) ? 1 : (MT * NT == 1 ? 4 : 2);
  f32x16 acc[MT][NT][KS];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int q = 0; q < KS; ++q)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][q][e] = 0.f;

  constexpr int AI = 2 * MT, BI = 2 * NT;
  float4 ra[AI], rb[BI];
  const int lrow16 = lane >> 2, lq = lane & 3;
  const i32x4 dy_rsrc = make_rsrc(p.dy, p.dy_bytes);
  const i32x4 x_rsrc = make_rsrc(p.x, p.x_bytes);
  // row state of this lane: row m = pbeg + (4 it + wave) * 16 + lrow16 in step `it`
  const unsigned m0 = (unsigned)(pbeg + wave * KR + lrow16);
  unsigned dyo = (m0 * (unsigned)p.dy_ld + (unsigned)(co0 + lq * 4)) * 4u;
  const unsigned dy_step = (unsigned)(4 * KR * p.dy_ld) * 4u;
  unsigned xo = 0, pp = 0, fb = 0;
  unsigned rm_off[BI];
  if (MODE == 2) {
    xo = (m0 * (unsigned)p.x_ld + (unsigned)(ci0 + lq * 4)) * 4u;
  } else {
    const uint32_t n = fdiv(m0, p.div_ohw);
    pp = m0 - n * p.div_ohw.d;
    fb = n * p.frame_bytes + (ROWMODE ? 0u : (unsigned)(ci0 + lq * 4) * 4u);
    if (ROWMODE) {
      // this lane's columns are fixed: their (run, offset) byte displacements from the pixel's first run, once
#pragma unroll
      for (int k = 0; k < BI; ++k) {
        const int cc = ci0 + lq * 4 + 16 * k;
        const uint32_t f = (uint32_t)cc >> 2, t = fdiv(f, p.div_rl4);
        rm_off[k] = (t * (uint32_t)(p.W * p.cp) + (f - t * p.div_rl4.d) * 4u) * 4u;
      }
    }
  }
  const unsigned x_step = (unsigned)(4 * KR * p.x_ld) * 4u;
  const int tap_y = r - p.pad, tap_x = s - p.pad;
  const unsigned xld4 = (unsigned)p.x_ld * 4u;
#if TBN_DIAG
  bool x_ok_staged = false;   // validity of the x row in flight (cost probe below)
#endif
  auto load_tiles = [&]() {
#pragma unroll
    for (int k = 0; k < AI; ++k) ra[k] = buf_load4(dy_rsrc, dyo + 64u * k);
    dyo += dy_step;
    if (MODE == 2) {
#pragma unroll
      for (int k = 0; k < BI; ++k) rb[k] = buf_load4(x_rsrc, xo + 64u * k);
      xo += x_step;
    } else {
      const unsigned oy = p.mul_ow != 0u ? __umulhi(pp, p.mul_ow) : pp;   // (OW == 1: the magic number 2^32 does not fit)
      const unsigned ox = pp - __umul24(oy, (unsigned)p.OW);
      if (MODE == 0) {
        const unsigned iy = __umul24(oy, (unsigned)p.stride) + (unsigned)tap_y;   // wraps below zero: fails the unsigned compare
        const unsigned ix = __umul24(ox, (unsigned)p.stride) + (unsigned)tap_x;
        const bool ok = (iy < (unsigned)p.H) && (ix < (unsigned)p.W);
        const unsigned off = fb + __umul24(__umul24(iy, (unsigned)p.W) + ix, xld4);
        const unsigned voff = ok ? off : TBN_OOB;
#if TBN_DIAG
        x_ok_staged = ok;
#endif
#pragma unroll
        for (int k = 0; k < BI; ++k) rb[k] = buf_load4(x_rsrc, voff + 64u * k);
      } else {
        // no masks: the border is in the image; rows >= M start beyond the image extent -> zeros
        const unsigned off = fb + __umul24(oy, p.row_step) + __umul24(ox, p.col_step);
#pragma unroll
        for (int k = 0; k < BI; ++k) rb[k] = buf_load4(x_rsrc, off + rm_off[k]);
      }
      // 64 rows on: in-frame index and frame base (64 = q64 * OH*OW + r64)
      pp += p.r64;
      const bool wrap = pp >= p.div_ohw.d;
      pp = wrap ? pp - p.div_ohw.d : pp;
      fb += wrap ? p.fb_hi : p.fb_lo;
    }
  };
#if TBN_DIAG
  // cost probe (see ConvP::fold_scale): BN apply + ReLU of the producer layer on the staged x rows; out-of-image taps stay zero
  float4 fsc[BI], fsh[BI];
  if (MODE == 0 && p.fold_scale != nullptr) {
#pragma unroll
    for (int k = 0; k < BI; ++k) {
      const int cc = min(ci0 + lq * 4 + 16 * k, p.Cin - 4);
      fsc[k] = *reinterpret_cast<const float4*>(p.fold_scale + cc);
      fsh[k] = *reinterpret_cast<const float4*>(p.fold_shift + cc);
    }
  }
#endif
  auto store_tiles = [&]() {
#if TBN_DIAG
    if (MODE == 0 && p.fold_scale != nullptr) {
#pragma unroll
      for (int k = 0; k < BI; ++k) {
        rb[k].x = x_ok_staged ? fmaxf(fmaf(rb[k].x, fsc[k].x, fsh[k].x), 0.f) : 0.f;
        rb[k].y = x_ok_staged ? fmaxf(fmaf(rb[k].y, fsc[k].y, fsh[k].y), 0.f) : 0.f;
        rb[k].z = x_ok_staged ? fmaxf(fmaf(rb[k].z, fsc[k].z, fsh[k].z), 0.f) : 0.f;
        rb[k].w = x_ok_staged ? fmaxf(fmaf(rb[k].w, fsc[k].w, fsh[k].w), 0.f) : 0.f;
      }
    }
#endif
#pragma unroll
    for (int k = 0; k < AI; ++k) *reinterpret_cast<float4*>(&At[lrow16 * WA + lq * 4 + 16 * k]) = ra[k];
#pragma unroll
    for (int k = 0; k < BI; ++k) *reinterpret_cast<float4*>(&Bt[lrow16 * WB + lq * 4 + 16 * k]) = rb[k];
  };

  // The LDS tiles are wave-private and a wave's LDS operations execute in program order, so the main loop
  // needs NO workgroup barrier: the four waves run decoupled (wave_barrier only pins the compiler's order).
  const int nsteps = (p.rows_per_split + 4 * KR - 1) / (4 * KR);
  const int lrow = lane & 31, lhalf = lane >> 5;
  load_tiles();
  for (int it = 0; it < nsteps; ++it) {
    if (!WABL(2) || it == 0) store_tiles();
    __builtin_amdgcn_wave_barrier();
    if (it + 1 < nsteps && !WABL(1)) load_tiles();
    if (WABL(4)) continue;
    // Fragments of k-pair kp+1 are read while k-pair kp multiplies (two register sets; the sched_group_barrier
    // chain pins the DS-read group / MFMA group alternation).  Consecutive MFMAs rotate over FOUR independent
    // accumulators: a dependent MFMA is only free right behind its producer or >= 4 MFMAs later -- alternating
    // two accumulators halves the rate (scripts/ubench, conv_igemm_kernel) -- so tiles with fewer than four
    // 32x32 sub-tiles split K over KS accumulators per sub-tile (summed before the cross-wave reduction).
    float a[2][MT], bb[2][NT];
    auto frag = [&](int buf, int kp) {
#pragma unroll
      for (int i = 0; i < MT; ++i) a[buf][i] = At[(2 * kp + lhalf) * WA + i * 32 + lrow];
#pragma unroll
      for (int j = 0; j < NT; ++j) bb[buf][j] = Bt[(2 * kp + lhalf) * WB + j * 32 + lrow];
    };
    frag(0, 0);
#pragma unroll
    for (int kp = 0; kp < KR / 2; ++kp) {
      if (kp + 1 < KR / 2) frag((kp + 1) & 1, kp + 1);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[i][j][kp % KS] =
              __builtin_amdgcn_mfma_f32_32x32x2f32(a[kp & 1][i], bb[kp & 1][j], acc[i][j][kp % KS], 0, 0, 0);
    }
    {
      constexpr int NM = MT * NT;
      constexpr int NRD = (MT + 1) / 2 + (NT + 1) / 2;   // b32 fragment reads are merged pairwise into ds_read2_b32
      __builtin_amdgcn_sched_group_barrier(0x100, 2 * NRD, 0);
#pragma unroll
      for (int g = 0; g < KR / 2 - 2; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, NM, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, NRD, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 2 * NM, 0);
    }
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();  // the reduction buffer below overlays every wave's tiles

  // cross-wave reduction, one 32x32 sub-tile at a time: red[wave][32*32]
  float* red = lds;
  float* obase = p.out + (size_t)split * p.Cout * p.K;
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = 8 * (e >> 2) + 4 * lhalf + (e & 3);
        float v = acc[i][j][0][e];
#pragma unroll
        for (int q = 1; q < KS; ++q) v += acc[i][j][q][e];
        red[wave * 1024 + row * 32 + lrow] = v;
      }
      __syncthreads();
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int el = tid + 256 * t;
        const float v = (red[el] + red[1024 + el]) + (red[2048 + el] + red[3072 + el]);
        const int row = el >> 5, col = el & 31;
        const int co = co0 + i * 32 + row, ci = ci0 + j * 32 + col;
        if (co < p.Cout && ci < p.Cin) obase[(size_t)co * p.K + tap * p.Cin + ci] = v;
      }
      __syncthreads();
    }
}

// Sums the split-K slabs of a weight gradient.  256 threads = (256 / SG) float4 outputs x SG slab lanes: lane g adds
// slabs g, g+SG, ... (two independent chains), the SG partial sums are combined through LDS in fixed order
// (deterministic).  SG = 16 for many slabs of a small matrix (the 7x7 stem: ~110 slabs of 57 KB) so that the
// grid still covers the chip; SG = 4 otherwise.
template <int SG>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                            int n4, int splits, size_t slab4) {
  constexpr int NO = 256 / SG;
  __shared__ float4 red[256];
  const int o = threadIdx.x % NO, g = threadIdx.x / NO;
  const int i = blockIdx.x * NO + o;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
  if (i < n4) {
    int s = g;
    for (; s + SG < splits; s += 2 * SG) {
      const float4 u = tbn_ld4<(TBN_BN_NT & 8) != 0>(part + 4 * ((size_t)s * slab4 + i)), v = tbn_ld4<(TBN_BN_NT & 8) != 0>(part + 4 * ((size_t)(s + SG) * slab4 + i));
      a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
      b.x += v.x; b.y += v.y; b.z += v.z; b.w += v.w;
    }
    if (s < splits) {
      const float4 u = tbn_ld4<(TBN_BN_NT & 8) != 0>(part + 4 * ((size_t)s * slab4 + i));
      a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
    }
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
  }
  red[threadIdx.x] = a;
  __syncthreads();
  if (g == 0 && i < n4) {
#pragma unroll
    for (int k = 1; k < SG; ++k) {
      const float4 v = red[k * NO + o];
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    reinterpret_cast<float4*>(out)[i] = a;
  }
}

// Wt[ci][T-1-tap][co] = W[co][tap][ci]  (tap-flipped, channel-transposed weights for dgrad)
__global__ void weight_flip_transpose_kernel(const float* __restrict__ w, float* __restrict__ wt, int Cout, int taps,
                                             int Cin) {
  __shared__ float tile[32][33];
  const int tap = blockIdx.z;
  const int co0 = blockIdx.y * 32, ci0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: 32 x 8
  for (int i = ty; i < 32; i += 8) {
    const int co = co0 + i, ci = ci0 + tx;
    tile[i][tx] = (co < Cout && ci < Cin) ? w[((size_t)co * taps + tap) * Cin + ci] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int ci = ci0 + i, co = co0 + tx;
    if (ci < Cin && co < Cout) wt[((size_t)ci * taps + (taps - 1 - tap)) * Cout + co] = tile[tx][i];
  }
}

// ------------------------------------------------------------------------------------------ host
// `rd` = the rider riding in this launch (rd.span == 0: none), `grid` = GEMM workgroups + rd.span (tbn_rider_place)
template <int MT, int NT, bool RM, int EPI, bool RED = false>
static void launch_conv_e(const ConvP& p, const RiderP& rd, int grid, hipStream_t st) {
  if (p.stages == 2)
    TBN_LAUNCH((conv_igemm_kernel<MT, NT, RM, EPI, 2, RED>), dim3(grid), dim3(256), 0, st, p, rd);
  else
    TBN_LAUNCH((conv_igemm_kernel<MT, NT, RM, EPI, 1, RED>), dim3(grid), dim3(256), 0, st, p, rd);
}
template <int MT, int NT, bool RM>
static void launch_conv(const ConvP& p, const RiderP& rd, int grid, hipStream_t st) {
  const bool scatter = (p.out_sy != 1) || (p.out_sx != 1);
  if (p.mode == CONV_EPI_STATS)
    launch_conv_e<MT, NT, RM, 1>(p, rd, grid, st);
  else if (p.mode == CONV_EPI_EVAL)
    launch_conv_e<MT, NT, RM, 2>(p, rd, grid, st);
  else if (!RM && scatter)
    launch_conv_e<MT, NT, false, 3>(p, rd, grid, st);
  else if (!RM && p.nred > 0)
    launch_conv_e<MT, NT, false, 0, true>(p, rd, grid, st);
  else
    launch_conv_e<MT, NT, RM, 0>(p, rd, grid, st);
}

// LDS bytes of the halo kernel; 0 if the shape is not a 3x3 / stride 1 / pad 1 layer it handles
size_t tbn_conv_halo_lds_bytes(const ConvP& p, int mt, int nt) {
  if (p.R != 3 || p.S != 3 || p.stride != 1 || p.pad != 1 || p.up != 1 || p.OH != p.H || p.OW != p.W || p.W > 64) return 0;
  return (size_t)(128 * mt + 2 * p.W + 3 + 2 * 32 * nt) * LDT * sizeof(float);
}

template <int MT, int NT, int EPI, bool RED>
static int launch_halo_e(const ConvP& p, const RiderP& rd, int grid, size_t lds_bytes, hipStream_t st) {
  static size_t allowed = 64 * 1024;   // per instantiation: raise the dynamic-LDS limit once when a shape needs it
  if (lds_bytes > allowed) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_halo_kernel<MT, NT, EPI, RED>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
      tbn_set_error("conv_halo: cannot raise the dynamic LDS limit");
      return TBN_ERR_LAUNCH;
    }
    allowed = 160 * 1024;
  }
  TBN_LAUNCH((conv_halo_kernel<MT, NT, EPI, RED>), dim3(grid), dim3(256), lds_bytes, st, p, rd);
  return TBN_OK;
}
template <int MT, int NT>
static int launch_halo(const ConvP& p, const RiderP& rd, int grid, size_t lds_bytes, hipStream_t st) {
  if (p.mode == CONV_EPI_STATS) return launch_halo_e<MT, NT, 1, false>(p, rd, grid, lds_bytes, st);
  if (p.mode == CONV_EPI_EVAL) return launch_halo_e<MT, NT, 2, false>(p, rd, grid, lds_bytes, st);
  if (p.nred > 0) return launch_halo_e<MT, NT, 0, true>(p, rd, grid, lds_bytes, st);
  return launch_halo_e<MT, NT, 0, false>(p, rd, grid, lds_bytes, st);
}

template <int MT, int NT>
static void launch_sk4(const ConvP& p, const RiderP& rd, int nblocks, hipStream_t st) {
  const dim3 grid(nblocks);
  if (p.mode == CONV_EPI_STATS)
    TBN_LAUNCH((conv_sk4_kernel<MT, NT, 1, false>), grid, dim3(256), 0, st, p, rd);
  else if (p.mode == CONV_EPI_EVAL)
    TBN_LAUNCH((conv_sk4_kernel<MT, NT, 2, false>), grid, dim3(256), 0, st, p, rd);
  else if (p.nred > 0)
    TBN_LAUNCH((conv_sk4_kernel<MT, NT, 0, true>), grid, dim3(256), 0, st, p, rd);
  else
    TBN_LAUNCH((conv_sk4_kernel<MT, NT, 0, false>), grid, dim3(256), 0, st, p, rd);
}

template <int MT, int NT>
static void launch_dma(const ConvP& p, const RiderP& rd, int nblocks, hipStream_t st) {
  const dim3 grid(nblocks);
  if (p.mode == CONV_EPI_STATS)
    TBN_LAUNCH((conv_dma_kernel<MT, NT, 1, false>), grid, dim3(256), 0, st, p, rd);
  else if (p.mode == CONV_EPI_EVAL)
    TBN_LAUNCH((conv_dma_kernel<MT, NT, 2, false>), grid, dim3(256), 0, st, p, rd);
  else if (p.nred > 0)
    TBN_LAUNCH((conv_dma_kernel<MT, NT, 0, true>), grid, dim3(256), 0, st, p, rd);
  else
    TBN_LAUNCH((conv_dma_kernel<MT, NT, 0, false>), grid, dim3(256), 0, st, p, rd);
}

void tbn_conv_pick_tile(int M, int Cout, int K, int* mt_out, int* nt_out) {
  double best = 1e300;
  int bm = 1, bn = 1;
  for (int mt = 1; mt <= 2; ++mt)
    for (int nt = 1; nt <= 4; ++nt) {
      const int tiles_m = cdiv(M, 128 * mt), tiles_n = cdiv(Cout, 32 * nt);
      const double blocks = (double)tiles_m * tiles_n;
      const double rounds = (double)((long)((blocks + 255) / 256));
      // MFMA cycles per k-step per wave + fixed per-step (barriers, exposed latency) + per-block cost
      const double per_block = (K / 32.0) * (mt * nt * 1024.0 + 350.0) + 4000.0;
      // wasted columns in the last N tile still cost load/issue slots but no MFMA (skipped)
      double cost = rounds * per_block;
      // prefer bigger tiles on ties (less L2 traffic)
      cost *= 1.0 + 0.01 / (mt * nt);
      if (cost < best) {
        best = cost;
        bm = mt;
        bn = nt;
      }
    }
  *mt_out = bm;
  *nt_out = bn;
}

// number of partial rows (M tiles) a data-gradient launch with a fused BN-backward reduce writes per reduce segment
int tbn_conv_red_rows(int N, int OH, int OW, int up, int tile_rows) {
  if (up == 1) return cdiv(N * OH * OW, tile_rows);
  int rows = 0;
  for (int py = 0; py < 2; ++py)
    for (int px = 0; px < 2; ++px) {
      const int ohs = (OH - py + 1) / 2, ows = (OW - px + 1) / 2;
      if (ohs > 0 && ows > 0) rows += cdiv(N * ohs * ows, tile_rows);
    }
  return rows;
}

// algorithmic HBM bytes of one conv launch: the input read once, the weights once, the output written once (read
// once more when accumulating; + the BN input read by a fused backward reduce)
static double conv_alg_bytes(const ConvP& p, int rowmode) {
  const double in = (double)p.N * p.H * p.W * (rowmode ? p.cp : p.Cin);
  double out = (double)p.M * p.Cout;
  if (p.flags & CONV_FLAG_ACCUM) out *= 2.0;
  if (p.nred > 0) out += (double)p.M * p.Cout;
  return 4.0 * (in + (double)p.Cout * p.Krow + out);
}

static int launch_conv_tiles(ConvP& p, int rowmode, int mt, int nt, hipStream_t st, const RiderP* rider) {
  static thread_local RiderP rd;      // by value into the kernel arguments (zeroed: no rider)
  if (rider != nullptr)
    rd = *rider;
  else
    memset(&rd, 0, sizeof(rd));
  if ((mt <= 0 || nt <= 0) && !rowmode && p.out_sy == 1 && p.out_sx == 1) {
    // heuristic launches (the head Linear layers: M = 96 ... 768 rows): a grid of 128-row tiles that leaves most CUs
    // idle takes the split-K tile kernel instead (never with partial-sum epilogues: their row count is the caller's)
    if (p.halo == 0 && p.mode != CONV_EPI_STATS && p.nred == 0 && p.K >= 256 &&
        cdiv(p.M, 128) * cdiv(p.Cout, 32) <= 128)
      p.halo = 3;
    if (p.halo == 3) mt = nt = 1;
  }
  if (mt <= 0 || nt <= 0) tbn_conv_pick_tile(p.M, p.Cout, p.K, &mt, &nt);
  if (p.stages != 1 && p.stages != 2) p.stages = (mt == 1) ? 2 : 1;  // big tiles: keep 2 workgroups per CU
  p.tiles_m = cdiv(p.M, 128 * mt);
  p.tiles_n = cdiv(p.Cout, 32 * nt);
  if (p.halo == 3 && !rowmode) {   // small-M layers: 32-row tiles, the four waves split K
    const bool scatter = (p.out_sy != 1) || (p.out_sx != 1);
    TBN_REQUIRE(!scatter && mt <= 2 && nt <= 2, "conv: the split-K tile kernel does not handle this launch");
    p.tiles_m = cdiv(p.M, 32 * mt);
    const int grid = tbn_rider_place(&rd, p.tiles_m * p.tiles_n);
    char nm[64];
    const int epi = p.mode == CONV_EPI_STATS ? 1 : (p.mode == CONV_EPI_EVAL ? 2 : 0);
    snprintf(nm, sizeof(nm), "conv_sk4_kernel<%d, %d, %d%s>", mt, nt, epi, (p.nred > 0 && epi == 0) ? ", true" : "");
    tbn_prof_begin(nm, p.alg_flops, st, conv_alg_bytes(p, rowmode));
#define TBN_SCASE(MTv, NTv) \
  if (mt == MTv && nt == NTv) launch_sk4<MTv, NTv>(p, rd, grid, st);
    TBN_SCASE(1, 1) TBN_SCASE(1, 2) TBN_SCASE(2, 1) TBN_SCASE(2, 2)
#undef TBN_SCASE
    tbn_prof_end(st);
    TBN_CHECK_LAUNCH("conv_sk4");
    return TBN_OK;
  }
  if (p.halo == 2 && !rowmode) {   // LDS-DMA staging
    const bool scatter = (p.out_sy != 1) || (p.out_sx != 1);
    TBN_REQUIRE(!scatter && mt <= 2 && nt <= 4, "conv: the LDS-DMA kernel does not handle this launch");
    const int grid = tbn_rider_place(&rd, p.tiles_m * p.tiles_n);
    char nm[64];
    const int epi = p.mode == CONV_EPI_STATS ? 1 : (p.mode == CONV_EPI_EVAL ? 2 : 0);
    snprintf(nm, sizeof(nm), "conv_dma_kernel<%d, %d, %d%s>", mt, nt, epi, (p.nred > 0 && epi == 0) ? ", true" : "");
    tbn_prof_begin(nm, p.alg_flops, st, conv_alg_bytes(p, rowmode));
#define TBN_DCASE(MTv, NTv) \
  if (mt == MTv && nt == NTv) launch_dma<MTv, NTv>(p, rd, grid, st);
    TBN_DCASE(1, 1) TBN_DCASE(1, 2) TBN_DCASE(1, 3) TBN_DCASE(1, 4) TBN_DCASE(2, 1) TBN_DCASE(2, 2) TBN_DCASE(2, 3) TBN_DCASE(2, 4)
#undef TBN_DCASE
    tbn_prof_end(st);
    TBN_CHECK_LAUNCH("conv_dma");
    return TBN_OK;
  }
  if (p.halo == 1 && !rowmode) {
    const size_t lds_bytes = tbn_conv_halo_lds_bytes(p, mt, nt);
    TBN_REQUIRE(lds_bytes > 0 && lds_bytes <= 160 * 1024, "conv: the LDS-halo kernel does not handle this shape / tile");
    TBN_REQUIRE(mt <= 2 && nt <= 4, "conv: unsupported halo tile %dx%d", mt, nt);
    const int grid = tbn_rider_place(&rd, p.tiles_m * p.tiles_n);
    char nm[64];
    const int epi = p.mode == CONV_EPI_STATS ? 1 : (p.mode == CONV_EPI_EVAL ? 2 : 0);
    snprintf(nm, sizeof(nm), "conv_halo_kernel<%d, %d, %d%s>", mt, nt, epi, (p.nred > 0 && epi == 0) ? ", true" : "");
    tbn_prof_begin(nm, p.alg_flops, st, conv_alg_bytes(p, rowmode));
    int rc = TBN_OK;
#define TBN_HCASE(MTv, NTv) \
  if (mt == MTv && nt == NTv) rc = launch_halo<MTv, NTv>(p, rd, grid, lds_bytes, st);
    TBN_HCASE(1, 1) TBN_HCASE(1, 2) TBN_HCASE(1, 3) TBN_HCASE(1, 4) TBN_HCASE(2, 1) TBN_HCASE(2, 2) TBN_HCASE(2, 3) TBN_HCASE(2, 4)
#undef TBN_HCASE
    tbn_prof_end(st);
    if (rc != TBN_OK) return rc;
    TBN_CHECK_LAUNCH("conv_halo");
    return TBN_OK;
  }
  const int grid = tbn_rider_place(&rd, p.tiles_m * p.tiles_n);
  {
    char nm[64];
    const int epi = p.mode == CONV_EPI_STATS ? 1 : (p.mode == CONV_EPI_EVAL ? 2 : ((p.out_sy != 1 || p.out_sx != 1) ? 3 : 0));
    snprintf(nm, sizeof(nm), "conv_igemm_kernel<%d, %d, %s, %d, %d%s>", mt, nt, rowmode ? "true" : "false", epi, p.stages,
             (p.nred > 0 && epi == 0 && !rowmode) ? ", true" : "");
    tbn_prof_begin(nm, p.alg_flops, st, conv_alg_bytes(p, rowmode));
  }
#define TBN_CASE(MTv, NTv)                                     \
  if (mt == MTv && nt == NTv) {                                \
    if (rowmode)                                               \
      launch_conv<MTv, NTv, true>(p, rd, grid, st);            \
    else                                                       \
      launch_conv<MTv, NTv, false>(p, rd, grid, st);           \
  } else
  TBN_CASE(1, 1) TBN_CASE(1, 2) TBN_CASE(1, 3) TBN_CASE(1, 4) TBN_CASE(2, 1) TBN_CASE(2, 2) TBN_CASE(2, 3)
  TBN_CASE(2, 4) {
    tbn_set_error("conv: unsupported tile %dx%d", mt, nt);
    return TBN_ERR_UNSUPPORTED;
  }
#undef TBN_CASE
  tbn_prof_end(st);
  TBN_CHECK_LAUNCH("conv_igemm");
  return TBN_OK;
}

// Fills the derived geometry and launches.  `up == 2` (data gradient of a stride-2 conv; the caller
// passes stride 1, pad = k-1-pad_fwd and tap-flipped weights) is decomposed into the 4 output-parity
// phases: each phase only visits the taps that hit a real (non zero-inserted) dy sample, so no MFMA
// work is spent on inserted zeros (2.25 instead of 9 taps per output pixel for 3x3).
// validation + derived fields common to every launch form; `*single` = 1 when the launch is a single GEMM
// (up == 1: geometry completely filled), 0 when the caller still has to split it into parity phases (up == 2)
static int conv_prepare(ConvP& p, int rowmode, int* single) {
  if (rowmode) {
    // R contiguous runs of Cin floats per output pixel, read from a physically zero-padded image without masks
    // K = the R runs, padded to a multiple of 32 with the first floats of a further run (zero weights; the caller's
    // image holds that row)
    const int kruns = cdiv(cdiv(p.R * p.Cin, 32) * 32, p.Cin);   // runs the padded K touches
    p.K = cdiv(p.R * p.Cin, 32) * 32;
    TBN_REQUIRE(p.Cin % 4 == 0 && p.cp >= 1 && p.pad == 0 && p.up == 1,
                "conv: packed-row mode needs run lengths in multiples of 4 floats, pad 0 (K = %d)", p.K);
    TBN_REQUIRE((p.OH - 1) * p.stride + kruns <= p.H && (p.OW - 1) * p.stride + cdiv(p.Cin, p.cp) <= p.W,
                "conv: packed-row mode reads beyond the padded image (%dx%d)", p.H, p.W);
  } else {
    TBN_REQUIRE(p.K % 32 == 0 && p.Cin % 32 == 0, "conv: K (%d) and per-tap Cin (%d) must be multiples of 32", p.K,
                p.Cin);
    TBN_REQUIRE(p.R <= 3 && p.S <= 3, "conv: filters larger than 3x3 take the packed-row (stem) path");
  }
  TBN_REQUIRE((rowmode || p.in_ld % 4 == 0) && p.nseg >= 1 && p.nseg <= TBN_CONV_MAXSEG, "conv: bad in_ld %d / nseg %d", p.in_ld,
              p.nseg);
  TBN_REQUIRE(p.nred >= 0 && p.nred <= TBN_CONV_MAXSEG && (p.nred == 0 || (p.mode == CONV_EPI_PLAIN && !rowmode)),
              "conv: the fused BN-backward reduce belongs to a plain (data-gradient) epilogue");
  for (int i = 0; i < p.nred; ++i) {
    TBN_REQUIRE(p.red[i].col_begin % 32 == 0 && (i == 0 ? p.red[i].col_begin == 0 : p.red[i].col_begin >= p.red[i - 1].col_begin + p.red[i - 1].C),
                "conv: reduce segment %d starts at column %d (x32, ascending, disjoint)", i, p.red[i].col_begin);
    if (p.red[i].y != nullptr) {
      TBN_REQUIRE(p.red[i].partial && p.red[i].stats && p.red[i].C > 0, "conv: reduce segment %d incomplete", i);
      const size_t yb = (((size_t)p.N * p.OH * p.OW - 1) * p.red[i].y_ld + p.red[i].C) * sizeof(float);
      TBN_REQUIRE(yb < (1ull << 31), "conv: reduce segment extent %zu B >= 2 GiB", yb);
      p.red[i].y_bytes = (unsigned)yb;
    }
  }
  TBN_REQUIRE(p.up == 1 || p.up == 2, "conv: up must be 1 or 2");
  if (p.flags & CONV_FLAG_HALO) p.halo = 1;
  if (p.flags & CONV_FLAG_DMA) p.halo = 2;
  if (p.flags & CONV_FLAG_SK4) p.halo = 3;
  TBN_REQUIRE(p.M > 0, "conv: empty problem");
  const size_t in_bytes = (size_t)p.N * p.H * p.W * (rowmode ? p.cp : p.in_ld) * sizeof(float);
  TBN_REQUIRE(in_bytes < (1ull << 31), "conv: input extent %zu B >= 2 GiB (process the frames in chunks)", in_bytes);
  p.in_bytes = (unsigned)in_bytes;
  if (p.alg_flops <= 0.0) p.alg_flops = 2.0 * p.M * (double)p.Cout * p.K;
  for (int i = 0; i < p.nseg; ++i) {
    const int cols = (i + 1 < p.nseg ? p.seg[i + 1].col_begin : p.Cout) - p.seg[i].col_begin;
    const size_t ob = (((size_t)p.N * p.OH * p.OW - 1) * p.seg[i].ld + cols) * sizeof(float);
    TBN_REQUIRE(ob < (1ull << 31), "conv: output extent %zu B >= 2 GiB (process the frames in chunks)", ob);
    // the epilogue picks the destination per 32-column sub-tile: segments must start on 32-column boundaries
    TBN_REQUIRE(i == 0 || p.seg[i].col_begin % 32 == 0, "conv: segment %d starts at column %d (must be a multiple of 32)",
                i, p.seg[i].col_begin);
    p.seg_bytes[i] = (unsigned)ob;
  }
  const int taps_full = rowmode ? 1 : p.R * p.S;
  p.Krow = rowmode ? p.K : taps_full * p.Cin;
  p.wt_bytes = (unsigned)((size_t)p.Cout * p.Krow * sizeof(float));
  p.div_rl4 = make_fastdiv((uint32_t)(rowmode ? p.Cin / 4 : 1));
  if (p.up == 1) {
    TBN_REQUIRE(taps_full <= 9, "conv: at most 9 taps (3x3)");
    p.OHs = p.OH;
    p.OWs = p.OW;
    p.out_sy = p.out_sx = 1;
    p.out_oy = p.out_ox = 0;
    p.in_sy = p.in_sx = p.stride;
    p.ntaps = taps_full;
    for (int t = 0; t < taps_full; ++t) {
      const int r = rowmode ? 0 : t / p.S, s = rowmode ? 0 : t % p.S;
      p.tap_dy[t] = r - p.pad;
      p.tap_dx[t] = s - p.pad;
      p.tap_koff[t] = t * p.Cin;
      p.tap_off[t] = rowmode ? 0 : (p.tap_dy[t] * p.W + p.tap_dx[t]) * p.in_ld * 4;
    }
    if (!rowmode) p.K = p.ntaps * p.Cin;
    p.ty0 = -p.pad;
    p.tny = rowmode ? 1 : p.R;
    p.tx0 = -p.pad;
    p.tnx = rowmode ? 1 : p.S;
    p.div_ohw = make_fastdiv((uint32_t)(p.OHs * p.OWs));
    p.div_ow = make_fastdiv((uint32_t)p.OWs);
    *single = 1;
    return TBN_OK;
  }
  *single = 0;
  return TBN_OK;
}

int tbn_launch_conv(ConvP p, int rowmode, int mt, int nt, hipStream_t st, const RiderP* rider) {
  int single = 0;
  const int prc = conv_prepare(p, rowmode, &single);
  if (prc != TBN_OK) return prc;
  if (single) return launch_conv_tiles(p, rowmode, mt, nt, st, rider);
  TBN_REQUIRE(!rowmode && p.stride == 1 && p.R == p.S && p.R * p.S <= 9, "conv: unsupported strided data gradient");
  const double flops_total = p.alg_flops;
  const int full_M = p.N * p.OH * p.OW;
  {
    // the scatter epilogue forms its byte offsets as 24-bit products (output pixel) x (row pitch in bytes): v_mul_u32_u24
    // silently drops the high bits, so both factors are checked here (the 2-GiB extent checks alone do not imply it
    // for narrow pitches)
    bool pitches_ok = true;
    for (int i = 0; i < p.nseg; ++i) pitches_ok = pitches_ok && (size_t)p.seg[i].ld * 4 < (1u << 24);
    for (int i = 0; i < p.nred; ++i) pitches_ok = pitches_ok && (size_t)p.red[i].y_ld * 4 < (1u << 24);
    TBN_REQUIRE(full_M < (1 << 24) && pitches_ok,
                "conv: strided data gradient of %d output pixels / a pitch beyond the 24-bit scatter arithmetic (chunk the frames)",
                full_M);
  }
  static thread_local ConvPhases phases;   // ~2 KB kernel argument, built in place
  phases.n = 0;
  bool empty_phase = false;
  for (int py = 0; py < 2; ++py)
    for (int px = 0; px < 2; ++px) {
      ConvP q = p;
      q.OHs = (p.OH - py + 1) / 2;
      q.OWs = (p.OW - px + 1) / 2;
      if (q.OHs <= 0 || q.OWs <= 0) continue;
      q.out_sy = q.out_sx = 2;
      q.out_oy = py;
      q.out_ox = px;
      q.in_sy = q.in_sx = 1;
      q.ntaps = 0;
      q.tny = q.tnx = 0;
      q.ty0 = q.tx0 = 1 << 20;
      for (int r = 0; r < p.R; ++r)
        if (((py - p.pad + r) & 1) == 0) {
          ++q.tny;
          if ((py - p.pad + r) / 2 < q.ty0) q.ty0 = (py - p.pad + r) / 2;
        }
      for (int s2 = 0; s2 < p.S; ++s2)
        if (((px - p.pad + s2) & 1) == 0) {
          ++q.tnx;
          if ((px - p.pad + s2) / 2 < q.tx0) q.tx0 = (px - p.pad + s2) / 2;
        }
      q.div_ohw = make_fastdiv((uint32_t)(q.OHs * q.OWs));
      q.div_ow = make_fastdiv((uint32_t)q.OWs);
      for (int r = 0; r < p.R; ++r) {
        if (((py - p.pad + r) & 1) != 0) continue;
        for (int s2 = 0; s2 < p.S; ++s2) {
          if (((px - p.pad + s2) & 1) != 0) continue;
          // floor division by 2 of a possibly negative even number
          q.tap_dy[q.ntaps] = (py - p.pad + r) / 2;
          q.tap_dx[q.ntaps] = (px - p.pad + s2) / 2;
          q.tap_koff[q.ntaps] = (r * p.S + s2) * p.Cin;
          q.tap_off[q.ntaps] = (q.tap_dy[q.ntaps] * p.W + q.tap_dx[q.ntaps]) * p.in_ld * 4;
          ++q.ntaps;
        }
      }
      q.M = p.N * q.OHs * q.OWs;
      q.alg_flops = flops_total * ((double)q.M / full_M);
      if (q.ntaps == 0) {
        // no contributing tap (1x1 stride-2 only): the gradient at this parity is zero
        empty_phase = true;
        continue;
      }
      q.K = q.ntaps * p.Cin;
      phases.ph[phases.n++] = q;
    }
  if (empty_phase && !(p.flags & CONV_FLAG_ACCUM)) {
    // pixels no phase writes: clear the whole destination first (stream order keeps this ahead of the phases)
    TBN_REQUIRE(p.nseg == 1, "conv: 1x1 strided data gradient writes one segment");
    if (hipMemset2DAsync(p.seg[0].ptr, (size_t)p.seg[0].ld * sizeof(float), 0, (size_t)p.Cout * sizeof(float),
                         (size_t)full_M, st) != hipSuccess) {
      tbn_set_error("conv: hipMemset2DAsync failed");
      return TBN_ERR_LAUNCH;
    }
  }
  if (phases.n == 0) return TBN_OK;
  // longest workgroups first: a phase's K is (its taps) x Cin -- 1, 2, 2 and 4 taps for a 3x3 / stride-2 layer -- and the
  // dispatcher hands out workgroups in block order, so with the 4-tap phase LAST (parity order) its 4x longer workgroups
  // started when the others were nearly done and ran the launch's tail alone
  static const int lpt = tbn_env_int("TBN_LPT", 1, 0, 1);   // A/B runs: 0 = parity order
  for (int a = 0; a < phases.n && lpt; ++a)
    for (int b = a + 1; b < phases.n; ++b)
      if (phases.ph[b].K > phases.ph[a].K) {
        const ConvP t = phases.ph[a];
        phases.ph[a] = phases.ph[b];
        phases.ph[b] = t;
      }
  // one tile shape for all phases (picked on the largest one), one launch
  int pmt = mt, pnt = nt;
  if (pmt <= 0 || pnt <= 0) {
    int big = 0;
    for (int i = 1; i < phases.n; ++i)
      if (phases.ph[i].M > phases.ph[big].M) big = i;
    tbn_conv_pick_tile(phases.ph[big].M, p.Cout, phases.ph[big].K, &pmt, &pnt);
  }
  const int stages = (p.stages == 1 || p.stages == 2) ? p.stages : ((pmt == 1) ? 2 : 1);
  phases.blk0[0] = 0;
  int red_rows = p.red_row0;
  for (int i = 0; i < phases.n; ++i) {
    ConvP& q = phases.ph[i];
    q.stages = stages;
    q.tiles_m = cdiv(q.M, 128 * pmt);
    q.tiles_n = cdiv(q.Cout, 32 * pnt);
    q.red_row0 = red_rows;      // the phases of one layer append their M tiles to the same partial buffers
    red_rows += q.tiles_m;
    phases.blk0[i + 1] = phases.blk0[i] + q.tiles_m * q.tiles_n;
  }
  {
    char nm[64];
    snprintf(nm, sizeof(nm), "conv_igemm_phases_kernel<%d, %d, %d%s>", pmt, pnt, stages, p.nred > 0 ? ", true" : "");
    tbn_prof_begin(nm, flops_total, st, conv_alg_bytes(p, 0));
  }
  static thread_local RiderP prd;
  if (rider != nullptr)
    prd = *rider;
  else
    memset(&prd, 0, sizeof(prd));
  const int pgrid = tbn_rider_place(&prd, phases.blk0[phases.n]);
#define TBN_PLAUNCH(MTv, NTv, STv, REDv) \
  TBN_LAUNCH((conv_igemm_phases_kernel<MTv, NTv, STv, REDv>), dim3(pgrid), dim3(256), 0, st, phases, prd)
#define TBN_PCASE(MTv, NTv)                          \
  if (pmt == MTv && pnt == NTv) {                    \
    if (stages == 2) {                               \
      if (p.nred > 0)                                \
        TBN_PLAUNCH(MTv, NTv, 2, true);              \
      else                                           \
        TBN_PLAUNCH(MTv, NTv, 2, false);             \
    } else {                                         \
      if (p.nred > 0)                                \
        TBN_PLAUNCH(MTv, NTv, 1, true);              \
      else                                           \
        TBN_PLAUNCH(MTv, NTv, 1, false);             \
    }                                                \
  } else
  TBN_PCASE(1, 1) TBN_PCASE(1, 2) TBN_PCASE(1, 3) TBN_PCASE(1, 4) TBN_PCASE(2, 1) TBN_PCASE(2, 2) TBN_PCASE(2, 3)
  TBN_PCASE(2, 4) {
    tbn_set_error("conv: unsupported tile %dx%d", pmt, pnt);
    return TBN_ERR_UNSUPPORTED;
  }
#undef TBN_PCASE
#undef TBN_PLAUNCH
  tbn_prof_end(st);
  TBN_CHECK_LAUNCH("conv_igemm_phases");
  return TBN_OK;
}

// ------------------------------------------------------------------------------------------
// Two INDEPENDENT convolutions in one launch: the 3x3 and double_3x3_1 layers of an inception block read different
// inputs and write different outputs, each alone leaves the chip partly idle on the 14x14 / 7x7 maps (259 .. 735
// workgroups on 256 CUs) and pays its own launch ramp and last-round tail.  One grid holds the tiles of both; a
// workgroup finds its member by a scalar compare.  Members share the tile shape, the kernel variant and the epilogue.
struct ConvPair {
  ConvP m[2];
  int blk1;   // first workgroup of member 1
};
template <int MT, int NT, int EPI, int STAGES, bool RED>
__global__ __launch_bounds__(256) void conv_pair_igemm_kernel(ConvPair q, RiderP rider) {
  __shared__ __attribute__((aligned(16))) float lds[STAGES * (128 * MT + 32 * NT) * LDT];
  TBN_RIDER_DISPATCH(rider, bid)
  const int mi = bid >= q.blk1 ? 1 : 0;
  conv_igemm_body<MT, NT, false, EPI, STAGES, RED>(q.m[mi], bid - (mi ? q.blk1 : 0), lds);
}
template <int MT, int NT, int EPI, bool RED>
__global__ __launch_bounds__(256) void conv_pair_halo_kernel(ConvPair q, RiderP rider) {
  extern __shared__ __attribute__((aligned(16))) float dyn_lds[];
  TBN_RIDER_DISPATCH(rider, bid)
  const int mi = bid >= q.blk1 ? 1 : 0;
  conv_halo_body<MT, NT, EPI, RED>(q.m[mi], bid - (mi ? q.blk1 : 0), dyn_lds);
}

template <int MT, int NT, int EPI, bool RED>
static int launch_pair_v(const ConvPair& q, const RiderP& rd, int blocks, int variant, size_t lds_bytes, hipStream_t st) {
  if (variant == 0) {   // LDS-halo members
    static size_t allowed = 64 * 1024;
    if (lds_bytes > allowed) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_pair_halo_kernel<MT, NT, EPI, RED>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
        tbn_set_error("conv_pair: cannot raise the dynamic LDS limit");
        return TBN_ERR_LAUNCH;
      }
      allowed = 160 * 1024;
    }
    TBN_LAUNCH((conv_pair_halo_kernel<MT, NT, EPI, RED>), dim3(blocks), dim3(256), lds_bytes, st, q, rd);
  } else if (variant == 2) {
    TBN_LAUNCH((conv_pair_igemm_kernel<MT, NT, EPI, 2, RED>), dim3(blocks), dim3(256), 0, st, q, rd);
  } else {
    TBN_LAUNCH((conv_pair_igemm_kernel<MT, NT, EPI, 1, RED>), dim3(blocks), dim3(256), 0, st, q, rd);
  }
  return TBN_OK;
}
template <int MT, int NT>
static int launch_pair(const ConvPair& q, const RiderP& rd, int blocks, int variant, size_t lds_bytes, hipStream_t st) {
  const ConvP& p = q.m[0];
  if (p.mode == CONV_EPI_STATS) return launch_pair_v<MT, NT, 1, false>(q, rd, blocks, variant, lds_bytes, st);
  if (p.mode == CONV_EPI_EVAL) return launch_pair_v<MT, NT, 2, false>(q, rd, blocks, variant, lds_bytes, st);
  if (p.nred > 0) return launch_pair_v<MT, NT, 0, true>(q, rd, blocks, variant, lds_bytes, st);
  return launch_pair_v<MT, NT, 0, false>(q, rd, blocks, variant, lds_bytes, st);
}

// variant: 0 LDS-halo (both members 3x3 / stride 1), 1 / 2 register-staged generic kernel with 1 / 2 LDS stages.
// Tiles: (1,1) (1,2) (2,1) (2,2).  Both members: unit-stride launches (no parity phases), same epilogue mode, reduce
// segments on both or on neither.
int tbn_launch_conv_pair(ConvP a, ConvP b, int variant, int mt, int nt, hipStream_t st, const RiderP* rider) {
  static thread_local ConvPair q;
  static thread_local RiderP rd;
  if (rider != nullptr)
    rd = *rider;
  else
    memset(&rd, 0, sizeof(rd));
  int sa = 0, sb = 0;
  int rc = conv_prepare(a, 0, &sa);
  if (rc != TBN_OK) return rc;
  rc = conv_prepare(b, 0, &sb);
  if (rc != TBN_OK) return rc;
  TBN_REQUIRE(sa && sb, "conv_pair: members must be unit-stride launches (no parity phases)");
  TBN_REQUIRE(a.mode == b.mode && (a.nred > 0) == (b.nred > 0) && a.flags == b.flags,
              "conv_pair: members must share the epilogue");
  TBN_REQUIRE(variant >= 0 && variant <= 2 && mt >= 1 && mt <= 2 && nt >= 1 && nt <= 2, "conv_pair: unsupported variant / tile");
  size_t lds_bytes = 0;
  if (variant == 0) {
    const size_t la = tbn_conv_halo_lds_bytes(a, mt, nt), lb = tbn_conv_halo_lds_bytes(b, mt, nt);
    TBN_REQUIRE(la > 0 && lb > 0 && la <= 160 * 1024 && lb <= 160 * 1024, "conv_pair: a member is not an LDS-halo shape");
    lds_bytes = la > lb ? la : lb;
  }
  // the member with the longer K loop first: its workgroups are the long ones, and the dispatcher hands out workgroups in
  // block order (a launch should not end with the long workgroups of the second member running alone)
  ConvP* ms[2] = {&a, &b};
  static const int lpt = tbn_env_int("TBN_LPT", 1, 0, 1);   // A/B runs: 0 = caller's order
  if (b.K > a.K && lpt) {
    ms[0] = &b;
    ms[1] = &a;
  }
  for (int i = 0; i < 2; ++i) {
    ms[i]->tiles_m = cdiv(ms[i]->M, 128 * mt);
    ms[i]->tiles_n = cdiv(ms[i]->Cout, 32 * nt);
    ms[i]->stages = variant == 2 ? 2 : 1;
    ms[i]->halo = variant == 0 ? 1 : 0;
    q.m[i] = *ms[i];
  }
  q.blk1 = ms[0]->tiles_m * ms[0]->tiles_n;
  const int blocks = tbn_rider_place(&rd, q.blk1 + ms[1]->tiles_m * ms[1]->tiles_n);
  {
    char nm[64];
    const int epi = a.mode == CONV_EPI_STATS ? 1 : (a.mode == CONV_EPI_EVAL ? 2 : 0);
    snprintf(nm, sizeof(nm), "conv_pair_%s_kernel<%d, %d, %d%s%s>", variant == 0 ? "halo" : "igemm", mt, nt, epi,
             variant == 2 ? ", 2" : (variant == 1 ? ", 1" : ""), (a.nred > 0 && epi == 0) ? ", true" : "");
    tbn_prof_begin(nm, a.alg_flops + b.alg_flops, st, conv_alg_bytes(a, 0) + conv_alg_bytes(b, 0));
  }
  rc = TBN_OK;
  if (mt == 1 && nt == 1) rc = launch_pair<1, 1>(q, rd, blocks, variant, lds_bytes, st);
  if (mt == 1 && nt == 2) rc = launch_pair<1, 2>(q, rd, blocks, variant, lds_bytes, st);
  if (mt == 2 && nt == 1) rc = launch_pair<2, 1>(q, rd, blocks, variant, lds_bytes, st);
  if (mt == 2 && nt == 2) rc = launch_pair<2, 2>(q, rd, blocks, variant, lds_bytes, st);
  tbn_prof_end(st);
  if (rc != TBN_OK) return rc;
  TBN_CHECK_LAUNCH("conv_pair");
  return TBN_OK;
}

template <int MT, int NT, int MODE>
static void launch_wgrad(const WgradP& p, int blocks, hipStream_t st) {
  // (round 6: an LDS-DMA form of this kernel -- dy / x rows global -> LDS without the VGPR round trip, two stages per wave,
  //  bit-identical results, commit 2773a1b -- measured 7 % SLOWER per launch, 499 -> 538 us on conv2_3x3: two stages of the
  //  64 x 64 tile need 74 KB of LDS = two workgroups per CU instead of four, and a DMA issued one step ahead has 0.85 us to land;
  //  profiles/r06_ab_wgrad_dma.txt.  Removed.)
  // experiment knob (A/B runs only): unused dynamic LDS caps the workgroups per CU (the 64 x 64 tile needs 64 registers and
  // 34 KB of LDS: four per CU; each split's 54 tile x tap workgroups share an x / dy slab through the XCD's 4-MB L2)
  // (clamped to what still launches: 64 KB of dynamic LDS minus the kernel's 34 KB of static LDS)
  static const int pad = tbn_env_int("TBN_WGRAD_LDS_PAD", 0, 0, 64 * 1024 - 36 * 1024);
  TBN_LAUNCH((conv_wgrad_kernel<MT, NT, MODE>), dim3(blocks), dim3(256), (MT == 2 && NT == 2) ? pad : 0, st, p);
}

// 32-column sub-tiles per workgroup along one dimension: 3 for 96 (and other odd multiples of 96), else 2 when the
// extent is a multiple of 64 or large, else 1
static int pick_wtile(int c) { return (c % 96 == 0 && c % 64 != 0) ? 3 : ((c % 64 == 0 || c > 96) ? 2 : 1); }

// plan: tile (0 = heuristic) + split-K.  The kernel's register budget admits `occ` workgroups per CU, so a grid runs
// in rounds of 256 * occ workgroups and a partly filled last round costs a full one: for every round count R = 1..8
// take the largest split count that still fits R rounds and keep the cheapest by
//   rounds * (K-steps per workgroup + 1) * step time  +  split-K reduce traffic
// (fitted to a sweep of 12 layer shapes x 24 plans, scripts/wgrad_ablate.py SWEEP=1: within 2 % of the best measured
// plan per layer, 11 % below the former "about 768 workgroups" rule on one stream).  Every split keeps >= 256 rows.
void tbn_wgrad_plan(int M, int Cout, int Cin, int taps, int* mt, int* nt, int* splits, int* rows_per_split) {
  if ((*mt < 1 || *mt > 3) && *mt != 5) {
    *mt = pick_wtile(Cout);
    // 160 output channels (3c / 4c 3x3 branches): 64-wide tiles multiply 17 % zero rows; one 160-wide tile (5 sub-tiles,
    // not the data-parallel stem path) fits exactly
    if (Cout == 160) *mt = 5;
  }
  if (*nt < 1 || *nt > 3) {
    *nt = pick_wtile(Cin);
    // 160 / 224 input channels: 64-wide tiles would multiply 20 / 14 % zero columns; 32-wide ones measured 5-22 %
    // faster there (profiles/r01_wgrad_tiles.txt).  The Cout side keeps its 64-wide tiles (narrow ones lose more
    // than the padding costs).
    if (*nt == 2 && Cin % 64 != 0 && cdiv(Cin, 64) * 64 * 100 >= Cin * 112) *nt = 1;
    // few rows, many input channels (the 1056-channel 1x1 group of 5a on the 7x7 maps): the 96-wide tile's single
    // workgroup per CU leaves 2.8 rounds of short loops; 64-wide tiles (3 % zero columns) run two per CU
    if (*nt == 3 && M <= 5120 && Cin >= 512) *nt = 2;   // (at 6144 rows, the 8x8 audio maps, the 96-wide tile still wins)
  }
  if (*mt == 5 && *nt == 3) *nt = 2;   // 5 x 3 sub-tiles do not fit the register file
  const int tiles = cdiv(Cout, 32 * *mt) * cdiv(Cin, 32 * *nt) * taps;
  const int max_splits = cdiv(M, 256);
  auto settle = [&](int s, int* rps) {
    if (s > max_splits) s = max_splits;
    if (s < 1) s = 1;
    *rps = cdiv(cdiv(M, s), 64) * 64;
    return cdiv(M, *rps);
  };
#if TBN_ABLATE
  if (const char* e = getenv("TBN_WGRAD_TARGET")) {   // scripts/wgrad_ablate.py: fixed grid target
    *splits = settle(cdiv(atoi(e), tiles), rows_per_split);
    return;
  }
#endif
  int occ = (*mt == 3 || *nt == 3 || (*mt == 5 && *nt > 1)) ? 1 : 2;   // workgroups per CU the cost model assumes
  {
    // experiment knob (A/B runs only): with VGPR-form accumulators the 64 x 64 and smaller tiles fit FOUR workgroups per CU
    static const int occ_env = tbn_env_int("TBN_WGRAD_OCC", 0, 0, 8);
    if (occ_env > 0 && occ == 2) occ = occ_env;
  }
  const int slots = 256 * occ;
  const double step_us = *mt * *nt * 8 * 64 / 2.4e3 * occ;  // 8 k-pairs x MT*NT MFMAs of 64 cycles, SIMD shared by occ waves
  const double slab_mb = (double)Cout * Cin * taps * 4e-6;
  double best = 0.0;
  for (int R = 1; R <= 8; ++R) {
    int rps;
    const int s = settle(R * slots / tiles, &rps);
    const double cost = cdiv(tiles * s, slots) * (rps / 64 + 1) * step_us + (s > 1 ? 0.3 * s * slab_mb : 0.0);
    if (R == 1 || cost < best) {
      best = cost;
      *splits = s;
      *rows_per_split = rps;
    }
  }
}

// split-K slab floats: the worst case over every tile the autotuner may choose
size_t tbn_wgrad_workspace_floats(int M, int Cout, int Cin, int taps) {
  size_t worst = 0;
  for (int mt = 1; mt <= 4; ++mt)
    for (int nt = 1; nt <= 3; ++nt) {
      int m = mt == 4 ? 5 : mt, n = nt, s, rps;
      tbn_wgrad_plan(M, Cout, Cin, taps, &m, &n, &s, &rps);
      const size_t need = s > 1 ? (size_t)s * Cout * taps * Cin : 0;
      if (need > worst) worst = need;
    }
  return worst;
}

int tbn_launch_wgrad(WgradP p, int rowmode, float* dw, float* workspace, hipStream_t st) {
  TBN_REQUIRE(p.Cin % 4 == 0 && p.Cout % 4 == 0 && p.dy_ld % 4 == 0 && (rowmode || p.x_ld % 4 == 0),
              "wgrad: channel counts / pitches must be multiples of 4");
  TBN_REQUIRE(p.M > 0, "wgrad: empty problem");
  {
    // exact extents: the kernel relies on the hardware range check for rows >= M and for ragged column tiles
    const size_t xb = rowmode ? (size_t)p.N * p.H * p.W * p.cp * sizeof(float)
                              : (((size_t)p.N * p.H * p.W - 1) * p.x_ld + p.Cin) * sizeof(float);
    const size_t db = (((size_t)p.M - 1) * p.dy_ld + p.Cout) * sizeof(float);
    TBN_REQUIRE(xb < (1ull << 31) && db < (1ull << 31), "wgrad: operand extent >= 2 GiB (process the frames in chunks)");
    p.x_bytes = (unsigned)xb;
    p.dy_bytes = (unsigned)db;
  }
#if TBN_ABLATE
  if (const char* e = getenv("TBN_WGRAD_ABLATE")) p.ablate = atoi(e);
  if (const char* e = getenv("TBN_WGRAD_MT")) p.mt = atoi(e);
  if (const char* e = getenv("TBN_WGRAD_NT")) p.nt = atoi(e);
#endif
  if (rowmode) {
    TBN_REQUIRE(p.taps == 1 && p.pad == 0 && p.rl > 0 && p.rl % 4 == 0 && p.cp >= 1 && p.Cin % p.rl == 0,
                "wgrad: packed-row mode needs taps 1, pad 0, Cin = R * rl, rl a multiple of 4");
    // (the last column tile may reach into one further run: finite image data into accumulator columns never stored)
    TBN_REQUIRE((p.OH - 1) * p.stride + p.Cin / p.rl <= p.H && (p.OW - 1) * p.stride + cdiv(p.rl, p.cp) <= p.W,
                "wgrad: packed-row mode reads beyond the padded image (%dx%d)", p.H, p.W);
  }
  p.div_rl4 = make_fastdiv((uint32_t)(rowmode ? p.rl / 4 : 1));
  int mt = p.mt, nt = p.nt, splits, rps;   // 0: heuristic tile
  tbn_wgrad_plan(p.M, p.Cout, p.Cin, p.taps, &mt, &nt, &splits, &rps);
  p.K = p.taps * p.Cin;
  p.tiles_co = cdiv(p.Cout, 32 * mt);
  p.tiles_ci = cdiv(p.Cin, 32 * nt);
  p.rows_per_split = rps;
  p.div_ohw = make_fastdiv((uint32_t)(p.OH * p.OW));
  p.div_ow = make_fastdiv((uint32_t)p.OW);
  // incremental row addressing (see the kernel): 64 rows per step
  TBN_REQUIRE(rps % 64 == 0, "wgrad: rows per split must be a multiple of 64");
  const int mode = rowmode ? 1 : ((p.taps == 1 && p.stride == 1 && p.pad == 0 && p.OH == p.H && p.OW == p.W) ? 2 : 0);
  {
    const uint64_t ohw = (uint64_t)p.OH * p.OW;
    TBN_REQUIRE(ohw * p.OW < (1ull << 32) && p.H < (1 << 23) && p.W < (1 << 23) && (uint64_t)p.H * p.W < (1ull << 24) &&
                    (uint64_t)p.x_ld * 4 < (1ull << 24),
                "wgrad: feature map too large for the 24-bit row arithmetic");
    p.mul_ow = p.OW == 1 ? 0u : (unsigned)(((1ull << 32) + p.OW - 1) / p.OW);   // 0: oy = pp (one-column maps)
    p.r64 = (unsigned)(64 % ohw);
    p.frame_bytes = (unsigned)((size_t)p.H * p.W * (rowmode ? p.cp : p.x_ld) * sizeof(float));
    p.fb_lo = (unsigned)(64 / ohw) * p.frame_bytes;
    p.fb_hi = p.fb_lo + p.frame_bytes;
    p.row_step = (unsigned)((size_t)p.stride * p.W * p.cp * sizeof(float));
    p.col_step = (unsigned)((size_t)p.stride * p.cp * sizeof(float));
    TBN_REQUIRE(!rowmode || (p.row_step < (1u << 24) && p.col_step < (1u << 24)), "wgrad: padded image row too long");
  }
  TBN_REQUIRE(splits == 1 || workspace != nullptr, "wgrad: split-K needs a workspace");
  p.out = splits > 1 ? workspace : dw;
  const int blocks = p.tiles_co * p.tiles_ci * p.taps * splits;
  if (p.alg_flops <= 0.0) p.alg_flops = 2.0 * p.M * (double)p.Cout * p.K;
  {
    char nm[64];
    snprintf(nm, sizeof(nm), "conv_wgrad_kernel<%d, %d, %d>", mt, nt, mode);
    // algorithmic bytes: dy and x read once, dW written once (split-K slabs are overhead, not counted)
    tbn_prof_begin(nm, p.alg_flops, st,
                   4.0 * ((double)p.M * p.Cout + (double)p.N * p.H * p.W * (rowmode ? p.cp : p.Cin) + (double)p.Cout * p.K));
  }
#define TBN_CASE(MTv, NTv)                                          \
  if (mt == MTv && nt == NTv) {                                     \
    if (mode == 1)                                                  \
      launch_wgrad<MTv, NTv, 1>(p, blocks, st);                     \
    else if (mode == 2)                                             \
      launch_wgrad<MTv, NTv, 2>(p, blocks, st);                     \
    else                                                            \
      launch_wgrad<MTv, NTv, 0>(p, blocks, st);                     \
  } else
  TBN_CASE(1, 1) TBN_CASE(1, 2) TBN_CASE(1, 3) TBN_CASE(2, 1) TBN_CASE(2, 2) TBN_CASE(2, 3) TBN_CASE(3, 1) TBN_CASE(3, 2)
  TBN_CASE(3, 3) TBN_CASE(5, 1) TBN_CASE(5, 2) {
    tbn_set_error("wgrad: unsupported tile");
    return TBN_ERR_UNSUPPORTED;
  }
#undef TBN_CASE
  tbn_prof_end(st);
  TBN_CHECK_LAUNCH("conv_wgrad");
  if (splits > 1) {
    const size_t n = (size_t)p.Cout * p.K;
    const int n4 = (int)(n / 4);
    if (splits >= 32 && n4 < 64 * 1024)
      TBN_KLAUNCH(splitk_reduce_kernel<16>, dim3(cdiv(n4, 16)), dim3(256), 0, st, workspace, dw, n4, splits, n / 4);
    else
      TBN_KLAUNCH(splitk_reduce_kernel<4>, dim3(cdiv(n4, 64)), dim3(256), 0, st, workspace, dw, n4, splits, n / 4);
    TBN_CHECK_LAUNCH("splitk_reduce");
  }
  return TBN_OK;
}

int tbn_launch_weight_flip_transpose(const float* w, float* wt, int Cout, int taps, int Cin, hipStream_t st) {
  TBN_KLAUNCH(weight_flip_transpose_kernel, dim3(cdiv(Cin, 32), cdiv(Cout, 32), taps), dim3(256), 0, st, w,
                     wt, Cout, taps, Cin);
  TBN_CHECK_LAUNCH("weight_flip_transpose");
  return TBN_OK;
}

// All layers of a backbone in ONE launch (the per-layer launches are ~4 us of pure latency each, 50 per
// backward pass): the layer table rides in the kernel arguments, a workgroup finds its layer by a scalar binary
// search (the former linear scan over up to 64 layers made this 10 M-float transpose take 218 us) and moves one
// 32 x 32 (co, ci) tile of every filter tap.
__global__ __launch_bounds__(256) void weight_flip_transpose_all_kernel(const float* __restrict__ w,
                                                                        float* __restrict__ wt, FlipTab tab) {
  // all taps of the tile are loaded first (up to 36 loads in flight per thread), ONE barrier, then all are written:
  // the former tap-by-tap form (two barriers per 4-KB tap slice) ran the 40 MB of a backbone at ~1 TB/s
  __shared__ float tile[9][32][33];   // indexed by tap < tab.taps[l] <= 9 (checked by the launcher)
  int lo = 0, hi = tab.n;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if ((int)blockIdx.x >= tab.blk0[mid])
      lo = mid;
    else
      hi = mid;
  }
  const int l = lo;
  const int Cout = tab.cout[l], Cin = tab.cin[l], taps = tab.taps[l];
  const int tci = (Cin + 31) >> 5;
  const int b = blockIdx.x - tab.blk0[l];
  const int ci0 = (b % tci) * 32, co0 = (b / tci) * 32;
  const float* wl = w + tab.w_off[l];
  float* wtl = wt + tab.w_off[l];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int tap = 0; tap < taps; ++tap)
#pragma unroll
    for (int i = ty; i < 32; i += 8) {
      const int co = co0 + i, ci = ci0 + tx;
      tile[tap][i][tx] = (co < Cout && ci < Cin) ? wl[((size_t)co * taps + tap) * Cin + ci] : 0.f;
    }
  __syncthreads();
  for (int tap = 0; tap < taps; ++tap)
#pragma unroll
    for (int i = ty; i < 32; i += 8) {
      const int ci = ci0 + i, co = co0 + tx;
      if (ci < Cin && co < Cout) wtl[((size_t)ci * taps + (taps - 1 - tap)) * Cout + co] = tile[tap][tx][i];
    }
}

int tbn_launch_weight_flip_transpose_all(const float* w, float* wt, const FlipTab& tab, hipStream_t st) {
  if (tab.n == 0) return TBN_OK;
  for (int l = 0; l < tab.n; ++l)   // the kernel stages every tap of a tile at once: tile[9][32][33]
    TBN_REQUIRE(tab.taps[l] >= 1 && tab.taps[l] <= 9, "weight_flip: layer %d has %d taps (at most 9 = 3x3)", l, (int)tab.taps[l]);
  TBN_KLAUNCH(weight_flip_transpose_all_kernel, dim3(tab.blk0[tab.n]), dim3(256), 0, st, w, wt, tab);
  TBN_CHECK_LAUNCH("weight_flip_transpose_all");
  return TBN_OK;
}
