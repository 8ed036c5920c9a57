// BN-Inception backbone engine: the whole per-modality conv stack as one natively scheduled
// program of HIP launches on one stream (forward train / forward eval / backward).
//
// Graph: reference core/models/bn_inception_audio.py:58-404 (layers) and :437-1003 (dataflow, concat
// order 1x1 | 3x3 | double-3x3 | pool) with the 7x7 stem of core/models/bn_inception.py:75-77.
// MI355X-first choices (vs. the reference's one nn.Module call per layer):
//   * NHWC activations; every branch writes straight into its channel slice of the block's
//     concat buffer (no torch.cat copies);
//   * the 1x1 convs that read the same block input (1x1, 3x3_reduce, double_3x3_reduce) run as ONE
//     GEMM (their parameters are adjacent in the flat parameter arrays), forward and backward;
//   * training-mode BatchNorm statistics are produced by the conv epilogue (no extra pass over y);
//   * all parameters of a backbone live in 6 flat arrays, gradients likewise -> the data-parallel
//     all-reduce and the optimizer touch a handful of large tensors (RCCL/xGMI friendly);
//   * the caller owns all memory: one workspace blob sized by tbn_backbone_workspace_bytes().
#include <string>
#include <vector>
#include <cstring>
#include <cstdio>

#include <cstdlib>
#include "tbn_common.h"
#include "tbn_kernels.h"
#include "../../include/tbn_hip.h"

namespace {

struct BlockSpec {
  const char* name;
  int cin, c1, c3r, c3, cdr, cd1, cd2;
  int pool;  // 0 avg->proj, 1 max(s1)->proj, 2 max(s2) pass-through
  int cp, stride;
};
const BlockSpec kBlocks[] = {
    {"3a", 192, 64, 64, 64, 64, 96, 96, 0, 32, 1},     {"3b", 256, 64, 64, 96, 64, 96, 96, 0, 64, 1},
    {"3c", 320, 0, 128, 160, 64, 96, 96, 2, 0, 2},     {"4a", 576, 224, 64, 96, 96, 128, 128, 0, 128, 1},
    {"4b", 576, 192, 96, 128, 96, 128, 128, 0, 128, 1}, {"4c", 576, 160, 128, 160, 128, 160, 160, 0, 128, 1},
    {"4d", 608, 96, 128, 192, 160, 192, 192, 0, 128, 1}, {"4e", 608, 0, 128, 192, 192, 256, 256, 2, 0, 2},
    {"5a", 1056, 352, 192, 320, 160, 224, 224, 0, 128, 1}, {"5b", 1024, 352, 192, 320, 192, 224, 224, 1, 128, 1},
};
const int kNumBlocks = 10;

struct Buf {
  int H, W, C;
  size_t off, doff;  // float offsets of z and dz in the workspace
};

struct Conv {
  int nparts;
  std::string names[3];
  int couts[3];
  int cin, cout, k, stride, pad;
  int inbuf, inH, inW, outH, outW;
  int dst_buf[3], dst_choff[3];
  size_t w_off, c_off;
  size_t y_off;
  int mt, nt, stages = 0;       // forward tile / LDS stages (0 = default)
  int d_mt, d_nt, d_stages = 0; // dgrad tile
  int w_mt = 0, w_nt = 0;       // wgrad tile (0 = heuristic)
  bool stem;
  bool dgrad_accum;    // dgrad adds into d(inbuf)
  bool need_dgrad;
  int fuse_pool = -1;  // training: index of the max pool that is the ONLY consumer of this conv's BN-ReLU output
  int group = -1, group_pos = 0, group_size = 1;  // training: BN kernels batched with the other members (bn_multi.hip)
  bool post_pool = false;  // 3x3 average pool between this 1x1 conv and its BN (see build_graph, pool_proj)
  size_t y2_off = 0;       // post_pool: pooled conv output = BN input (y_off holds the un-pooled conv output)
};

struct Pool {
  int kind;  // 0 avg3, 1 max
  int inbuf, outbuf, out_choff, stride, pad;
  size_t argmax_off;   // bytes
  bool bwd_accum;
  bool fused = false;  // training: executed inside its producer conv's BN apply / BN backward (see Conv::fuse_pool)
};

struct Op {
  int kind;  // 0 conv, 1 pool
  int idx;
};

int pool_out(int in, int k, int stride, int pad, bool ceil_mode) {
  int num = in + 2 * pad - k;
  int o = (ceil_mode ? (num + stride - 1) / stride : num / stride) + 1;
  if (ceil_mode && (o - 1) * stride >= in + pad) --o;
  return o;
}

}  // namespace

struct tbn_backbone_plan {
  int cin0, frames, H, W;
  int cp, kw;  // stem channel padding / padded filter-row length
  // stem conv as executed: 7 filter rows / stride 2 / pad 3 on the (H, W, cp) image, or -- one input channel, even
  // H and W -- 4 rows / stride 1 / pad 2 on the 2x2 space-to-depth image (H/2, W/2, 4), see pool.hip
  bool s2d;
  int stem_rows, stem_stride, stem_pad, stem_H, stem_W;
  std::vector<Buf> bufs;
  std::vector<Conv> convs;
  std::vector<Pool> pools;
  std::vector<Op> ops;
  int out_buf;
  size_t weight_floats, chan_floats;
  // workspace layout (float offsets unless noted)
  size_t x0_off, stats_off, partial_off, coef_off, wsplit_off, wt_off, wpack_off, dwpack_off;
  FlipTab flip;  // data-gradient weights: one flip/transpose launch per backward pass
  size_t partial_floats, wsplit_floats, wt_floats;
  size_t argmax_bytes_off, total_bytes_train, total_bytes_eval;
  size_t eval_floats;
  // fork/join events for the optional aux (weight-gradient) stream; created on first use
  hipEvent_t ev[8];
  int n_ev = 0;
};

namespace {

int add_buf(tbn_backbone_plan* P, int H, int W, int C) {
  Buf b;
  b.H = H;
  b.W = W;
  b.C = C;
  b.off = b.doff = 0;
  P->bufs.push_back(b);
  return (int)P->bufs.size() - 1;
}

int add_conv(tbn_backbone_plan* P, int nparts, const std::string* names, const int* couts, int cin, int k, int stride,
             int pad, int inbuf, const int* dst_buf, const int* dst_choff, bool stem) {
  Conv c;
  c.nparts = nparts;
  c.cout = 0;
  for (int i = 0; i < nparts; ++i) {
    c.names[i] = names[i];
    c.couts[i] = couts[i];
    c.dst_buf[i] = dst_buf[i];
    c.dst_choff[i] = dst_choff[i];
    c.cout += couts[i];
  }
  c.cin = cin;
  c.k = k;
  c.stride = stride;
  c.pad = pad;
  c.inbuf = inbuf;
  c.inH = P->bufs[inbuf].H;
  c.inW = P->bufs[inbuf].W;
  c.outH = (c.inH + 2 * pad - k) / stride + 1;
  c.outW = (c.inW + 2 * pad - k) / stride + 1;
  c.stem = stem;
  c.w_off = P->weight_floats;
  c.c_off = P->chan_floats;
  P->weight_floats += (size_t)c.cout * k * k * cin;
  P->chan_floats += c.cout;
  c.need_dgrad = !stem;
  c.dgrad_accum = false;
  c.y_off = 0;
  P->convs.push_back(c);
  Op o = {0, (int)P->convs.size() - 1};
  P->ops.push_back(o);
  return o.idx;
}

int add_pool(tbn_backbone_plan* P, int kind, int inbuf, int outbuf, int out_choff, int stride, int pad) {
  Pool p;
  p.kind = kind;
  p.inbuf = inbuf;
  p.outbuf = outbuf;
  p.out_choff = out_choff;
  p.stride = stride;
  p.pad = pad;
  p.argmax_off = 0;
  p.bwd_accum = false;
  P->pools.push_back(p);
  Op o = {1, (int)P->pools.size() - 1};
  P->ops.push_back(o);
  return o.idx;
}

// false: the input size makes the reference graph itself inconsistent (its torch.cat of a stride-2 conv branch and the
// ceil-mode pass-through max pool raises)
bool build_graph(tbn_backbone_plan* P) {
  const int cin0 = P->cin0;
  P->cp = (cin0 + 3) / 4 * 4;
  P->kw = (7 * P->cp + 31) / 32 * 32;
  P->s2d = cin0 == 1 && P->H % 2 == 0 && P->W % 2 == 0;
  P->stem_rows = P->s2d ? 4 : 7;
  P->stem_stride = P->s2d ? 1 : 2;
  P->stem_pad = P->s2d ? 2 : 3;
  P->stem_H = P->s2d ? P->H / 2 : P->H;
  P->stem_W = P->s2d ? P->W / 2 : P->W;
  if (P->s2d) P->kw = 32;   // 4 pixels x 4 parities = 16 real slots
  P->weight_floats = P->chan_floats = 0;
  const int x0 = add_buf(P, P->H, P->W, P->cp);   // logical extent; the s2d layout has the same number of floats
  // stem
  int h1 = (P->H + 6 - 7) / 2 + 1, w1 = (P->W + 6 - 7) / 2 + 1;
  const int c1 = add_buf(P, h1, w1, 64);
  {
    std::string n = "conv1_7x7_s2";
    int co = 64, db = c1, dc = 0;
    add_conv(P, 1, &n, &co, cin0, 7, 2, 3, x0, &db, &dc, true);
  }
  int hp = pool_out(h1, 3, 2, 0, true), wp = pool_out(w1, 3, 2, 0, true);
  const int p1 = add_buf(P, hp, wp, 64);
  add_pool(P, 1, c1, p1, 0, 2, 0);
  const int c2r = add_buf(P, hp, wp, 64);
  {
    std::string n = "conv2_3x3_reduce";
    int co = 64, db = c2r, dc = 0;
    add_conv(P, 1, &n, &co, 64, 1, 1, 0, p1, &db, &dc, false);
  }
  const int c2 = add_buf(P, hp, wp, 192);
  {
    std::string n = "conv2_3x3";
    int co = 192, db = c2, dc = 0;
    add_conv(P, 1, &n, &co, 64, 3, 1, 1, c2r, &db, &dc, false);
  }
  int h = pool_out(hp, 3, 2, 0, true), w = pool_out(wp, 3, 2, 0, true);
  int x = add_buf(P, h, w, 192);
  add_pool(P, 1, c2, x, 0, 2, 0);

  for (int bi = 0; bi < kNumBlocks; ++bi) {
    const BlockSpec& B = kBlocks[bi];
    const std::string pre = std::string("inception_") + B.name;
    int oh = h, ow = w;
    if (B.stride == 2) {
      oh = (h + 2 - 3) / 2 + 1;
      ow = (w + 2 - 3) / 2 + 1;
    }
    const int ctot = B.c1 + B.c3 + B.cd2 + (B.pool == 2 ? B.cin : B.cp);
    const int O = add_buf(P, oh, ow, ctot);
    const int T1 = add_buf(P, h, w, B.c3r);
    const int T2 = add_buf(P, h, w, B.cdr);
    const int T3 = add_buf(P, h, w, B.cd1);
    int pool_in = x;
    // fused 1x1 group on the block input
    {
      std::string names[3];
      int couts[3], db[3], dc[3], n = 0;
      if (B.c1) {
        names[n] = pre + "_1x1";
        couts[n] = B.c1;
        db[n] = O;
        dc[n] = 0;
        ++n;
      }
      names[n] = pre + "_3x3_reduce";
      couts[n] = B.c3r;
      db[n] = T1;
      dc[n] = 0;
      ++n;
      names[n] = pre + "_double_3x3_reduce";
      couts[n] = B.cdr;
      db[n] = T2;
      dc[n] = 0;
      ++n;
      add_conv(P, n, names, couts, B.cin, 1, 1, 0, x, db, dc, false);
    }
    // The 3x3, double_3x3_1 and pool_proj convs only depend on the fused 1x1 group: they are issued back to back
    // and share ONE batched BN finalize / apply (forward) and BN-backward reduce / finalize / apply launch set.
    int members[3], nm = 0;
    if (B.pool == 1) {   // 5b: 3x3 / stride-1 max pool of the block input feeds pool_proj (not linear: stays in front)
      const int XP = add_buf(P, h, w, B.cin);
      add_pool(P, 1, x, XP, 0, 1, 1);
      pool_in = XP;
    }
    {
      std::string n = pre + "_3x3";
      int co = B.c3, db = O, dc = B.c1;
      members[nm++] = add_conv(P, 1, &n, &co, B.c3r, 3, B.stride, 1, T1, &db, &dc, false);
    }
    {
      std::string n = pre + "_double_3x3_1";
      int co = B.cd1, db = T3, dc = 0;
      members[nm++] = add_conv(P, 1, &n, &co, B.cdr, 3, 1, 1, T2, &db, &dc, false);
    }
    if (B.pool == 2) {
      if (pool_out(h, 3, 2, 0, true) != oh || pool_out(w, 3, 2, 0, true) != ow) return false;
      add_pool(P, 1, x, O, B.c1 + B.c3 + B.cd2, 2, 0);
    } else if (B.pool == 0) {
      // reference: pool_proj(avg_pool3x3(x)).  A 3x3 / stride 1 / count_include_pad average and a (bias-free) 1x1
      // conv commute exactly, so the conv runs on the block input and the POOLING runs on its 32..128 output
      // channels instead of the 192..1056 input channels: 5-8x fewer bytes through the HBM-bound pool kernels
      // (forward and backward), no pooled copy of the block input.  The bias is added after the pool (it is folded
      // into the BN statistics / shift like everywhere else), which is where the reference adds it.
      std::string n = pre + "_pool_proj";
      int co = B.cp, db = O, dc = B.c1 + B.c3 + B.cd2;
      const int ci = add_conv(P, 1, &n, &co, B.cin, 1, 1, 0, x, &db, &dc, false);
      P->convs[ci].post_pool = true;
      members[nm++] = ci;
    } else {
      std::string n = pre + "_pool_proj";
      int co = B.cp, db = O, dc = B.c1 + B.c3 + B.cd2;
      members[nm++] = add_conv(P, 1, &n, &co, B.cin, 1, 1, 0, pool_in, &db, &dc, false);
    }
    for (int k = 0; k < nm; ++k) {
      Conv& c = P->convs[members[k]];
      c.group = bi;
      c.group_pos = k;
      c.group_size = nm;
    }
    {
      std::string n = pre + "_double_3x3_2";
      int co = B.cd2, db = O, dc = B.c1 + B.c3;
      add_conv(P, 1, &n, &co, B.cd1, 3, B.stride, 1, T3, &db, &dc, false);
    }
    x = O;
    h = oh;
    w = ow;
  }
  P->out_buf = x;

  // backward write order: walk ops in reverse, first writer of a d-buffer overwrites, later ones add
  std::vector<char> written(P->bufs.size(), 0);
  for (int i = (int)P->ops.size() - 1; i >= 0; --i) {
    const Op& o = P->ops[i];
    if (o.kind == 0) {
      Conv& c = P->convs[o.idx];
      if (c.need_dgrad) {
        c.dgrad_accum = written[c.inbuf] != 0;
        written[c.inbuf] = 1;
      }
    } else {
      Pool& p = P->pools[o.idx];
      p.bwd_accum = written[p.inbuf] != 0;
      written[p.inbuf] = 1;
    }
  }

  // conv -> BN -> ReLU -> max pool with no other reader of the BN output (the stem: conv1 -> pool1, conv2_3x3 -> pool2):
  // in training the pool runs inside the BN apply and its backward inside the BN backward, so the full-resolution z
  // and dz tensors (the largest of the network) are never written or re-read.
  std::vector<int> readers(P->bufs.size(), 0);
  for (auto& c : P->convs) ++readers[c.inbuf];
  for (auto& q : P->pools) ++readers[q.inbuf];
  for (size_t pi = 0; pi < P->pools.size(); ++pi) {
    Pool& q = P->pools[pi];
    if (q.kind != 1 || q.bwd_accum || readers[q.inbuf] != 1 || q.inbuf == P->out_buf) continue;
    for (auto& c : P->convs)
      if (c.nparts == 1 && c.dst_buf[0] == q.inbuf && c.dst_choff[0] == 0 && c.cout == P->bufs[q.inbuf].C) {
        c.fuse_pool = (int)pi;
        q.fused = true;
      }
  }
  return true;
}

// Timing diagnostics only (-DTBN_DIAG=1 build, never shipped): TBN_DIAG_SKIP=<bit mask> drops kernel groups
// (1 finalize, 2 bn_apply, 4 pools, 8 bn_bwd_reduce, 16 bn_bwd_apply) to measure what each costs the step.
#ifndef TBN_DIAG
#define TBN_DIAG 0
#endif
static inline bool diag_skip(int bit) {
#if TBN_DIAG
  static const int mask = getenv("TBN_DIAG_SKIP") ? atoi(getenv("TBN_DIAG_SKIP")) : 0;
  return (mask & bit) != 0;
#else
  (void)bit;
  return false;
#endif
}
void plan_memory(tbn_backbone_plan* P) {
  const size_t R = P->frames;
  size_t off = 0;
  auto take = [&](size_t n) {
    size_t o = off;
    off += (n + 63) / 64 * 64;  // 256-B aligned float regions
    return o;
  };
  for (auto& b : P->bufs) b.off = take(R * b.H * b.W * b.C);
  P->x0_off = P->bufs[0].off;
  P->stats_off = take(4 * P->chan_floats);  // mean | rstd | scale | shift
  // per-layer tiles and scratch sizes
  size_t partial = 0, wsplit = 0, wtf = 0;
  for (auto& c : P->convs) {
    const int M = (int)(R * c.outH * c.outW);
    const int K = c.stem ? P->stem_rows * P->kw : c.k * c.k * c.cin;
    tbn_conv_pick_tile(M, c.cout, K, &c.mt, &c.nt);
    size_t a = (size_t)cdiv(M, 128) * 2 * c.cout;  // worst case (mt = 1): autotune may pick any tile
    size_t bparts = (size_t)tbn_bn_bwd_parts(M, c.cout) * 2 * c.cout;
    if (a > partial) partial = a;
    if (bparts > partial) partial = bparts;
    const int taps = c.stem ? P->stem_rows : c.k * c.k, ci = c.stem ? P->kw : c.cin;
    size_t ws = tbn_wgrad_workspace_floats(M, c.cout, ci, taps);
    if (ws > wsplit) wsplit = ws;
    if (c.need_dgrad) {
      const int Md = (int)(R * c.inH * c.inW);
      tbn_conv_pick_tile(Md, c.cin, c.k * c.k * c.cout, &c.d_mt, &c.d_nt);
    }
  }
  // flipped / transposed copy of every data-gradient weight, at the layer's own offset
  wtf = P->weight_floats;
  memset(&P->flip, 0, sizeof(P->flip));
  for (auto& c : P->convs)
    if (c.need_dgrad && P->flip.n < 64) {
      FlipTab& f = P->flip;
      f.w_off[f.n] = (int)c.w_off;
      f.cout[f.n] = (short)c.cout;
      f.cin[f.n] = (short)c.cin;
      f.taps[f.n] = (short)(c.k * c.k);
      f.blk0[f.n + 1] = f.blk0[f.n] + cdiv(c.cin, 32) * cdiv(c.cout, 32) * c.k * c.k;
      ++f.n;
    }
  P->partial_floats = partial;
  P->wsplit_floats = wsplit;
  P->wt_floats = wtf;
  P->partial_off = take(TBN_BN_MAXL * partial);   // one region per member of a batched BN group
  P->wpack_off = take((size_t)64 * 7 * P->kw);
  for (auto& c : P->convs)   // conv output / pooled conv output of the pool-after-conv layers: needed in eval too
    if (c.post_pool) {
      c.y_off = take(R * c.outH * c.outW * c.cout);
      c.y2_off = take(R * c.outH * c.outW * c.cout);
    }
  P->eval_floats = off;
  // training-only regions
  for (auto& c : P->convs)
    if (!c.post_pool) c.y_off = take(R * c.outH * c.outW * c.cout);
  for (size_t i = 1; i < P->bufs.size(); ++i) {
    if ((int)i == P->out_buf) continue;  // gradient of the final feature map is supplied by the caller
    P->bufs[i].doff = take(R * P->bufs[i].H * P->bufs[i].W * P->bufs[i].C);
  }
  P->coef_off = take(TBN_BN_MAXL * 3 * 1024);
  P->wsplit_off = take(wsplit);
  P->wt_off = take(wtf);
  P->dwpack_off = take((size_t)64 * 7 * P->kw);
  size_t bytes = off * sizeof(float);
  P->argmax_bytes_off = bytes;
  for (auto& p : P->pools)
    if (p.kind == 1) {
      const Buf& ob = P->bufs[p.outbuf];
      const Buf& ib = P->bufs[p.inbuf];
      p.argmax_off = bytes;
      bytes += align_up(R * ob.H * ob.W * ib.C, 256);
    }
  P->total_bytes_train = bytes;
  P->total_bytes_eval = P->eval_floats * sizeof(float);
}

}  // namespace

extern "C" {

int tbn_backbone_plan_create(int in_channels, int frames, int height, int width, tbn_backbone_plan** out) {
  TBN_REQUIRE(out != nullptr, "plan_create: null out");
  TBN_REQUIRE(in_channels >= 1 && in_channels <= 16 && frames >= 1 && height >= 32 && width >= 32,
              "plan_create: unsupported shape (C=%d, frames=%d, %dx%d)", in_channels, frames, height, width);
  TBN_REQUIRE((long)frames * height * width < (1l << 31) / 4, "plan_create: too many pixels per call (chunk the batch)");
  tbn_backbone_plan* P = new tbn_backbone_plan();
  P->cin0 = in_channels;
  P->frames = frames;
  P->H = height;
  P->W = width;
  if (!build_graph(P) || P->bufs[P->out_buf].H < 1 || P->bufs[P->out_buf].W < 1) {
    delete P;
    tbn_set_error("plan_create: %dx%d input is not a valid BN-Inception size (the stride-2 conv branches and the "
                  "ceil-mode pass-through max pool of inception_3c / 4e disagree; the reference's torch.cat raises)",
                  height, width);
    return TBN_ERR_UNSUPPORTED;
  }
  {
    int nd = 0;
    for (auto& c : P->convs) nd += c.need_dgrad ? 1 : 0;
    if (nd > 64) {
      delete P;
      tbn_set_error("plan_create: %d data-gradient layers exceed the flip table (64)", nd);
      return TBN_ERR_UNSUPPORTED;
    }
  }
  plan_memory(P);
  *out = P;
  return TBN_OK;
}

void tbn_backbone_plan_destroy(tbn_backbone_plan* p) {
  if (!p) return;
  for (int i = 0; i < p->n_ev; ++i) (void)hipEventDestroy(p->ev[i]);
  delete p;
}

int tbn_backbone_num_convs(const tbn_backbone_plan* P) {
  int n = 0;
  for (auto& c : P->convs) n += c.nparts;
  return n;
}

int tbn_backbone_conv_info(const tbn_backbone_plan* P, int idx, tbn_conv_info* info) {
  TBN_REQUIRE(info != nullptr, "conv_info: null info");
  int n = 0;
  for (auto& c : P->convs) {
    size_t w = c.w_off, ch = c.c_off;
    for (int i = 0; i < c.nparts; ++i) {
      if (n == idx) {
        memset(info, 0, sizeof(*info));
        snprintf(info->name, sizeof(info->name), "%s", c.names[i].c_str());
        info->cin = c.cin;
        info->cout = c.couts[i];
        info->ksize = c.k;
        info->stride = c.stride;
        info->pad = c.pad;
        info->weight_offset = w;
        info->channel_offset = ch;
        return TBN_OK;
      }
      w += (size_t)c.couts[i] * c.k * c.k * c.cin;
      ch += c.couts[i];
      ++n;
    }
  }
  tbn_set_error("conv_info: index %d out of range", idx);
  return TBN_ERR_ARG;
}

// debugging / per-layer parity tests: where one conv's tensors live inside the workspace.
// kind 0: z = relu(bn(conv)) destination slice; 1: raw conv output y (after backward: dy);
// 2: gradient wrt z.  offset in floats (for kind 2 of the final block the gradient is external: -1).
int tbn_backbone_tensor_info(const tbn_backbone_plan* P, const char* conv_name, int kind, long* offset, int* rows,
                             int* cols, int* ld) {
  for (auto& c : P->convs) {
    int col = 0;
    for (int i = 0; i < c.nparts; ++i) {
      if (c.names[i] == conv_name) {
        const Buf& db = P->bufs[c.dst_buf[i]];
        *rows = P->frames * c.outH * c.outW;
        *cols = c.couts[i];
        if (kind == 1) {
          *offset = (long)((c.post_pool ? c.y2_off : c.y_off) + col);
          *ld = c.cout;
        } else if (kind == 0) {
          *offset = (long)(db.off + c.dst_choff[i]);
          *ld = db.C;
        } else {
          *offset = c.dst_buf[i] == P->out_buf ? -1 : (long)(db.doff + c.dst_choff[i]);
          *ld = db.C;
        }
        return TBN_OK;
      }
      col += c.couts[i];
    }
  }
  tbn_set_error("tensor_info: unknown conv '%s'", conv_name);
  return TBN_ERR_ARG;
}

size_t tbn_backbone_weight_floats(const tbn_backbone_plan* P) { return P->weight_floats; }
size_t tbn_backbone_channel_floats(const tbn_backbone_plan* P) { return P->chan_floats; }
size_t tbn_backbone_workspace_bytes(const tbn_backbone_plan* P, int training) {
  return training ? P->total_bytes_train : P->total_bytes_eval;
}
int tbn_backbone_out_shape(const tbn_backbone_plan* P, int* h, int* w, int* c) {
  const Buf& b = P->bufs[P->out_buf];
  if (h) *h = b.H;
  if (w) *w = b.W;
  if (c) *c = b.C;
  return TBN_OK;
}

#define TBN_TRY(expr)            \
  do {                           \
    int rc__ = (expr);           \
    if (rc__ != TBN_OK) return rc__; \
  } while (0)

int tbn_backbone_forward(const tbn_backbone_plan* P, int training, const float* x_nchw,
                         const tbn_backbone_params* prm, void* workspace, size_t workspace_bytes,
                         float** features_out, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  TBN_REQUIRE(P && x_nchw && prm && workspace && features_out, "backbone_forward: null argument");
  TBN_REQUIRE(workspace_bytes >= tbn_backbone_workspace_bytes(P, training), "backbone_forward: workspace too small");
  TBN_REQUIRE(((uintptr_t)workspace & 255) == 0, "backbone_forward: workspace must be 256-B aligned");
  float* ws = (float*)workspace;
  const int R = P->frames;
  float* mean = ws + P->stats_off;
  float* rstd = mean + P->chan_floats;
  float* scale = rstd + P->chan_floats;
  float* shift = scale + P->chan_floats;
  float* wpack = ws + P->wpack_off;

  if (P->s2d) {
    TBN_TRY(tbn_launch_nchw1_to_s2d(x_nchw, ws + P->x0_off, R, P->H, P->W, st));
    TBN_TRY(tbn_launch_pack_stem_weight_s2d(prm->weight + P->convs[0].w_off, wpack, 64, st));
  } else {
    TBN_TRY(tbn_launch_nchw_to_nhwc_pad(x_nchw, ws + P->x0_off, R, P->cin0, P->H, P->W, P->cp, st));
    TBN_TRY(tbn_launch_pack_stem_weight(prm->weight + P->convs[0].w_off, wpack, 64, P->cin0, P->cp, P->kw, st));
  }
  if (!training)
    TBN_TRY(tbn_launch_bn_fold(prm->gamma, prm->beta, prm->running_mean, prm->running_var, prm->bias, prm->eps, scale,
                               shift, (int)P->chan_floats, st));

  BnFwdBatch fwd_batch;
  memset(&fwd_batch, 0, sizeof(fwd_batch));
  for (const Op& o : P->ops) {
    if (o.kind == 0) {
      const Conv& c = P->convs[o.idx];
      const Buf& ib = P->bufs[c.inbuf];
      ConvP p;
      memset(&p, 0, sizeof(p));
      p.in = ws + ib.off;
      p.in_ld = ib.C;
      p.N = R;
      p.H = c.inH;
      p.W = c.inW;
      p.OH = c.outH;
      p.OW = c.outW;
      p.Cout = c.cout;
      p.stride = c.stride;
      p.pad = c.pad;
      p.up = 1;
      p.M = R * c.outH * c.outW;
      p.bias = nullptr;  // training: cancels in the batch-stat BN (finalize adds it to running_mean); eval: folded into shift
      p.alg_flops = 2.0 * p.M * (double)c.cout * c.k * c.k * c.cin;
      if (c.stem) {
        p.wt = wpack;
        p.Cin = P->kw;
        p.R = P->stem_rows;
        p.S = 1;
        p.K = P->stem_rows * P->kw;
        p.cp = P->cp;
        p.H = P->stem_H;          // s2d: the 4-row / stride-1 conv on the space-to-depth image
        p.W = P->stem_W;
        p.stride = P->stem_stride;
        p.pad = P->stem_pad;
      } else {
        p.wt = prm->weight + c.w_off;
        p.Cin = c.cin;
        p.R = p.S = c.k;
        p.K = c.k * c.k * c.cin;
      }
      Seg zs[3];
      int col = 0;
      for (int i = 0; i < c.nparts; ++i) {
        const Buf& db = P->bufs[c.dst_buf[i]];
        zs[i].ptr = ws + db.off + c.dst_choff[i];
        zs[i].ld = db.C;
        zs[i].col_begin = col;
        col += c.couts[i];
      }
      tbn_prof_label(("fwd " + c.names[c.nparts - 1]).c_str());
      // members of a batched BN group (training) write their statistics partials into their own scratch region
      const bool grouped = training && c.group >= 0;
      float* part = ws + P->partial_off + (grouped ? (size_t)c.group_pos * P->partial_floats : 0);
      float* bn_in = ws + c.y_off;     // BN input (post_pool: the pooled conv output)
      int nparts = 0;
      if (c.post_pool) {
        // bias-free 1x1 conv on the block input -> 3x3 average of its few output channels -> BN (+ReLU) -> concat
        float* u_raw = ws + c.y_off;
        bn_in = ws + c.y2_off;
        p.mode = CONV_EPI_PLAIN;
        p.nseg = 1;
        p.seg[0].ptr = u_raw;
        p.seg[0].ld = c.cout;
        p.seg[0].col_begin = 0;
        p.stages = c.stages;
        TBN_TRY(tbn_launch_conv(p, false, c.mt, c.nt, st));
        if (!diag_skip(4))
          TBN_TRY(tbn_launch_avgpool3_fwd(u_raw, c.cout, bn_in, c.cout, R, c.outH, c.outW, c.cout, 0, st));
        if (training) TBN_TRY(tbn_launch_bn_stats(bn_in, c.cout, p.M, c.cout, part, &nparts, st));
      } else if (training) {
        p.mode = CONV_EPI_STATS;
        p.nseg = 1;
        p.seg[0].ptr = bn_in;
        p.seg[0].ld = c.cout;
        p.seg[0].col_begin = 0;
        p.stat_partial = part;
        p.stages = c.stages;
        TBN_TRY(tbn_launch_conv(p, c.stem, c.mt, c.nt, st));
        nparts = cdiv(p.M, 128 * c.mt);
      }
      if (grouped) {
        BnFwdLayer& L = fwd_batch.l[c.group_pos];
        L.y = bn_in;
        L.P = p.M;
        L.C = c.cout;
        L.partial = part;
        L.nparts = nparts;
        L.gamma = prm->gamma + c.c_off;
        L.beta = prm->beta + c.c_off;
        L.conv_bias = prm->bias + c.c_off;
        L.running_mean = prm->running_mean + c.c_off;
        L.running_var = prm->running_var + c.c_off;
        L.save_mean = mean + c.c_off;
        L.save_rstd = rstd + c.c_off;
        L.scale = scale + c.c_off;
        L.shift = shift + c.c_off;
        L.nseg = c.nparts;
        for (int i = 0; i < c.nparts; ++i) L.seg[i] = zs[i];
        if (c.group_pos == c.group_size - 1) {   // last member issued: one finalize + one apply for the group
          fwd_batch.n = c.group_size;
          fwd_batch.momentum = prm->momentum;
          fwd_batch.eps = prm->eps;
          if (!diag_skip(3)) TBN_TRY(tbn_launch_bn_fwd_multi(fwd_batch, st));
        }
      } else if (training) {
        if (!diag_skip(1))
          TBN_TRY(tbn_launch_bn_finalize(part, nparts, p.M, c.cout, prm->gamma + c.c_off, prm->beta + c.c_off,
                                         prm->bias + c.c_off, prm->running_mean + c.c_off, prm->running_var + c.c_off,
                                         prm->momentum, prm->eps, mean + c.c_off, rstd + c.c_off, scale + c.c_off,
                                         shift + c.c_off, st));
        if (c.fuse_pool >= 0) {
          const Pool& q = P->pools[c.fuse_pool];
          const Buf& ob = P->bufs[q.outbuf];
          if (!diag_skip(2))
            TBN_TRY(tbn_launch_bn_apply_maxpool(bn_in, (int)R, c.outH, c.outW, c.cout, scale + c.c_off, shift + c.c_off,
                                                ws + ob.off + q.out_choff, ob.C, (uint8_t*)workspace + q.argmax_off,
                                                ob.H, ob.W, q.stride, q.pad, st));
        } else if (!diag_skip(2)) {
          TBN_TRY(tbn_launch_bn_apply(bn_in, p.M, c.cout, scale + c.c_off, shift + c.c_off, zs, c.nparts, st));
        }
      } else if (c.post_pool) {
        if (!diag_skip(2))   // eval: scale / shift are the folded running statistics (conv bias included)
          TBN_TRY(tbn_launch_bn_apply(bn_in, p.M, c.cout, scale + c.c_off, shift + c.c_off, zs, c.nparts, st));
      } else {
        p.mode = CONV_EPI_EVAL;
        p.scale = scale + c.c_off;
        p.shift = shift + c.c_off;
        p.nseg = c.nparts;
        for (int i = 0; i < c.nparts; ++i) p.seg[i] = zs[i];
        p.stages = c.stages;
        TBN_TRY(tbn_launch_conv(p, c.stem, c.mt, c.nt, st));
      }
    } else {
      const Pool& q = P->pools[o.idx];
      const Buf& ib = P->bufs[q.inbuf];
      const Buf& ob = P->bufs[q.outbuf];
      if (training && q.fused) continue;  // ran inside the producer's BN apply
      if (q.kind == 0) {
        if (!diag_skip(4)) TBN_TRY(tbn_launch_avgpool3_fwd(ws + ib.off, ib.C, ws + ob.off + q.out_choff, ob.C, R, ib.H, ib.W, ib.C, 0, st));
      } else {
        uint8_t* am = training ? (uint8_t*)workspace + q.argmax_off : nullptr;
        if (!diag_skip(4)) TBN_TRY(tbn_launch_maxpool_fwd(ws + ib.off, ib.C, ws + ob.off + q.out_choff, ob.C, am, R, ib.H, ib.W, ib.C,
                                       ob.H, ob.W, q.stride, q.pad, st));
      }
    }
  }
  *features_out = ws + P->bufs[P->out_buf].off;
  return TBN_OK;
}

// weight-gradient launch parameters of one conv (shared by backward and autotune)
static void fill_wgrad(const tbn_backbone_plan* P, const Conv& c, float* ws, int R, WgradP* wp) {
  const Buf& ib = P->bufs[c.inbuf];
  memset(wp, 0, sizeof(*wp));
  wp->dy = ws + c.y_off;
  wp->dy_ld = c.cout;
  wp->x = ws + ib.off;
  wp->x_ld = ib.C;
  wp->N = R;
  wp->H = c.inH;
  wp->W = c.inW;
  wp->OH = c.outH;
  wp->OW = c.outW;
  wp->Cout = c.cout;
  wp->stride = c.stride;
  wp->pad = c.pad;
  wp->M = R * c.outH * c.outW;
  wp->alg_flops = 2.0 * wp->M * (double)c.cout * c.k * c.k * c.cin;
  wp->mt = c.w_mt;
  wp->nt = c.w_nt;
  if (c.stem) {
    wp->Cin = P->kw;
    wp->R = P->stem_rows;
    wp->S = 1;
    wp->taps = P->stem_rows;
    wp->cp = P->cp;
    wp->H = P->stem_H;
    wp->W = P->stem_W;
    wp->stride = P->stem_stride;
    wp->pad = P->stem_pad;
  } else {
    wp->Cin = c.cin;
    wp->R = wp->S = c.k;
    wp->taps = c.k * c.k;
  }
}


// One-time tile autotuning: times every (MT, NT) tile of the forward and data-gradient implicit GEMM
// of each layer on the real shapes (2 launches each, hipEvents) and stores the fastest in the plan.
// Synchronises the stream (the only entry point that does); activations in `workspace` are clobbered.
int tbn_backbone_autotune(tbn_backbone_plan* P, int training, const tbn_backbone_params* prm, void* workspace,
                          size_t workspace_bytes, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  TBN_REQUIRE(P && prm && workspace, "autotune: null argument");
  TBN_REQUIRE(workspace_bytes >= tbn_backbone_workspace_bytes(P, training), "autotune: workspace too small");
  float* ws = (float*)workspace;
  const int R = P->frames;
  float* scale = ws + P->stats_off + 2 * P->chan_floats;
  float* shift = scale + P->chan_floats;
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
    tbn_set_error("autotune: hipEventCreate failed");
    return TBN_ERR_LAUNCH;
  }
  int rc = TBN_OK;
  for (auto& c : P->convs) {
    const Buf& ib = P->bufs[c.inbuf];
    for (int pass = 0; pass < 2 && rc == TBN_OK; ++pass) {  // 0: forward, 1: data gradient
      if (pass == 1 && (!training || !c.need_dgrad)) continue;
      ConvP p;
      memset(&p, 0, sizeof(p));
      p.N = R;
      p.up = 1;
      p.nseg = 1;
      if (pass == 0) {
        p.in = ws + ib.off;
        p.in_ld = ib.C;
        p.H = c.inH;
        p.W = c.inW;
        p.OH = c.outH;
        p.OW = c.outW;
        p.Cout = c.cout;
        p.stride = c.stride;
        p.pad = c.pad;
        p.M = R * c.outH * c.outW;
        p.bias = nullptr;
        if (c.stem) {
          p.wt = ws + P->wpack_off;
          p.Cin = P->kw;
          p.R = P->stem_rows;
          p.S = 1;
          p.K = P->stem_rows * P->kw;
          p.cp = P->cp;
          p.H = P->stem_H;          // s2d: the 4-row / stride-1 conv on the space-to-depth image
          p.W = P->stem_W;
          p.stride = P->stem_stride;
          p.pad = P->stem_pad;
        } else {
          p.wt = prm->weight + c.w_off;
          p.Cin = c.cin;
          p.R = p.S = c.k;
          p.K = c.k * c.k * c.cin;
        }
        p.mode = training ? CONV_EPI_STATS : CONV_EPI_EVAL;
        p.scale = scale + c.c_off;
        p.shift = shift + c.c_off;
        p.stat_partial = ws + P->partial_off;
        // write into the layer's own y buffer (training) or its first destination (eval)
        if (training) {
          p.seg[0].ptr = ws + c.y_off;
          p.seg[0].ld = c.cout;
        } else {
          const Buf& db = P->bufs[c.dst_buf[0]];
          p.seg[0].ptr = ws + db.off + c.dst_choff[0];
          p.seg[0].ld = db.C;
          p.Cout = c.couts[0];  // eval tuning on the first fused part only keeps all writes in range
          if (c.nparts > 1) continue;
        }
      } else {
        p.in = ws + c.y_off;
        p.in_ld = c.cout;
        p.wt = ws + P->wt_off + c.w_off;
        p.H = c.outH;
        p.W = c.outW;
        p.OH = c.inH;
        p.OW = c.inW;
        p.Cin = c.cout;
        p.Cout = c.cin;
        p.R = p.S = c.k;
        p.stride = 1;
        p.pad = c.k - 1 - c.pad;
        p.up = c.stride;      // stride-2 layers: the four parity phases in one launch
        p.M = R * c.inH * c.inW;
        p.K = c.k * c.k * c.cout;
        p.mode = CONV_EPI_PLAIN;
        p.seg[0].ptr = ws + ib.doff;
        p.seg[0].ld = ib.C;
      }
      // fastest of the candidates in isolation.  (Measured alternatives that lost: preferring the largest tile
      // within 3-15 % of the fastest -- monotonically slower steps; autotuning the weight-gradient tile the same
      // way -- 0.4 % slower than the size heuristic of tbn_wgrad_plan.)
      float best = 1e30f;
      int bm = 1, bn = 1, bs = 2;
      for (int mt = 1; mt <= 2 && rc == TBN_OK; ++mt)
        for (int nt = 1; nt <= 4 && rc == TBN_OK; ++nt)
          for (int stg = 1; stg <= 2 && rc == TBN_OK; ++stg) {
            if (32 * (nt - 1) >= p.Cout) continue;
            float ms = 0.f;
            p.stages = stg;
            for (int rep = 0; rep < 2 && rc == TBN_OK; ++rep) {
              (void)hipEventRecord(e0, st);
              rc = tbn_launch_conv(p, c.stem && pass == 0, mt, nt, st);
              (void)hipEventRecord(e1, st);
              (void)hipEventSynchronize(e1);
              (void)hipEventElapsedTime(&ms, e0, e1);
            }
            if (ms < best) {
              best = ms;
              bm = mt;
              bn = nt;
              bs = stg;
            }
          }
      if (pass == 0) {
        c.mt = bm;
        c.nt = bn;
        c.stages = bs;
      } else {
        c.d_mt = bm;
        c.d_nt = bn;
        c.d_stages = bs;
      }
    }
    if (rc != TBN_OK) break;
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return rc;
}

int tbn_backbone_backward(const tbn_backbone_plan* P, const float* dfeatures, const tbn_backbone_params* prm,
                          const tbn_backbone_grads* g, void* workspace, size_t workspace_bytes, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  hipStream_t aux = (hipStream_t)g->aux_stream;
  tbn_backbone_plan* PM = const_cast<tbn_backbone_plan*>(P);  // event pool only
  if (aux != nullptr && aux != st && PM->n_ev == 0) {
    for (int i = 0; i < 8; ++i) {
      if (hipEventCreateWithFlags(&PM->ev[i], hipEventDisableTiming) != hipSuccess) {
        tbn_set_error("backbone_backward: hipEventCreate failed");
        return TBN_ERR_LAUNCH;
      }
      PM->n_ev = i + 1;
    }
  }
  if (aux == st) aux = nullptr;
  int ev_next = 0;
  TBN_REQUIRE(P && dfeatures && prm && g && workspace, "backbone_backward: null argument");
  TBN_REQUIRE(g->dweight && g->dbias, "backbone_backward: dweight/dbias required");
  TBN_REQUIRE(workspace_bytes >= P->total_bytes_train, "backbone_backward: workspace too small");
  float* ws = (float*)workspace;
  const int R = P->frames;
  float* mean = ws + P->stats_off;
  float* rstd = mean + P->chan_floats;
  float* scale = rstd + P->chan_floats;
  float* shift = scale + P->chan_floats;
  float* partial = ws + P->partial_off;
  float* coef = ws + P->coef_off;
  auto dptr = [&](int buf) -> const float* {
    return buf == P->out_buf ? dfeatures : ws + P->bufs[buf].doff;
  };

  TBN_TRY(tbn_launch_weight_flip_transpose_all(prm->weight, ws + P->wt_off, P->flip, st));
  for (int oi = (int)P->ops.size() - 1; oi >= 0; --oi) {
    const Op& o = P->ops[oi];
    if (o.kind == 1) {
      const Pool& q = P->pools[o.idx];
      const Buf& ib = P->bufs[q.inbuf];
      const Buf& ob = P->bufs[q.outbuf];
      if (q.inbuf == 0) continue;
      if (q.fused) continue;  // its gradient gather runs inside the producer's BN backward
      float* din = ws + ib.doff;
      if (q.kind == 0) {
        if (!diag_skip(4)) TBN_TRY(tbn_launch_avgpool3_fwd(dptr(q.outbuf) + q.out_choff, ob.C, din, ib.C, R, ib.H, ib.W, ib.C, q.bwd_accum,
                                        st));
      } else {
        if (!diag_skip(4)) TBN_TRY(tbn_launch_maxpool_bwd(dptr(q.outbuf) + q.out_choff, ob.C, (const uint8_t*)workspace + q.argmax_off,
                                       din, ib.C, R, ib.H, ib.W, ib.C, ob.H, ob.W, q.stride, q.pad, q.bwd_accum, st));
      }
      continue;
    }
    const Conv& c = P->convs[o.idx];
    const Buf& ib = P->bufs[c.inbuf];
    const int M = R * c.outH * c.outW;
    float* const dconv = ws + c.y_off;                         // gradient of the conv output, read by wgrad / dgrad
    float* y = c.post_pool ? ws + c.y2_off : dconv;             // BN input; becomes dy in place
    CSeg dz[3];
    int col = 0;
    for (int i = 0; i < c.nparts; ++i) {
      const Buf& db = P->bufs[c.dst_buf[i]];
      dz[i].ptr = dptr(c.dst_buf[i]) + c.dst_choff[i];
      dz[i].ld = db.C;
      dz[i].col_begin = col;
      col += c.couts[i];
    }
    const bool bn_grad = g->dgamma && g->dbeta && (g->bn_grad_layers == 2 || (g->bn_grad_layers == 1 && o.idx == 0));
    if (c.group >= 0) {
      // batched BN backward of the whole group, issued when the reverse walk reaches its LAST member: the gradients
      // wrt every member's BN output exist by then (concat gradient, or the data gradient of double_3x3_2)
      if (c.group_pos == c.group_size - 1) {
        BnBwdBatch bb;
        memset(&bb, 0, sizeof(bb));
        bb.n = c.group_size;
        for (int k = 0; k < c.group_size; ++k) {
          const Conv& m = P->convs[o.idx - (c.group_size - 1 - k)];   // members are consecutive convs
          BnBwdLayer& L = bb.l[m.group_pos];
          int mc = 0;
          for (int i = 0; i < m.nparts; ++i) {
            const Buf& db = P->bufs[m.dst_buf[i]];
            L.dz[i].ptr = dptr(m.dst_buf[i]) + m.dst_choff[i];
            L.dz[i].ld = db.C;
            L.dz[i].col_begin = mc;
            mc += m.couts[i];
          }
          L.nseg = m.nparts;
          L.y = L.dy = m.post_pool ? ws + m.y2_off : ws + m.y_off;
          L.P = R * m.outH * m.outW;
          L.C = m.cout;
          L.scale = scale + m.c_off;
          L.shift = shift + m.c_off;
          L.mean = mean + m.c_off;
          L.rstd = rstd + m.c_off;
          L.partial = partial + (size_t)m.group_pos * P->partial_floats;
          L.coef = coef + (size_t)m.group_pos * 3 * 1024;
          const bool mg = g->dgamma && g->dbeta && g->bn_grad_layers == 2;   // members are never the first layer
          L.dgamma = mg ? g->dgamma + m.c_off : nullptr;
          L.dbeta = mg ? g->dbeta + m.c_off : nullptr;
          L.dbias = g->dbias + m.c_off;
        }
        if (!diag_skip(27)) TBN_TRY(tbn_launch_bn_bwd_multi(bb, st));
      }
    } else {
    if (c.fuse_pool >= 0) {
      const Pool& q = P->pools[c.fuse_pool];
      const Buf& ob = P->bufs[q.outbuf];
      const float* dpool = dptr(q.outbuf) + q.out_choff;
      const uint8_t* am = (const uint8_t*)workspace + q.argmax_off;
      if (!diag_skip(8))
        TBN_TRY(tbn_launch_bn_bwd_reduce_pooled(dpool, ob.C, am, R, c.outH, c.outW, ob.H, ob.W, q.stride, q.pad, y,
                                                c.cout, scale + c.c_off, shift + c.c_off, mean + c.c_off,
                                                rstd + c.c_off, partial, st));
    } else if (!diag_skip(8)) {
      TBN_TRY(tbn_launch_bn_bwd_reduce(dz, c.nparts, y, M, c.cout, scale + c.c_off, shift + c.c_off, mean + c.c_off,
                                       rstd + c.c_off, partial, st));
    }
    if (!diag_skip(1))
      TBN_TRY(tbn_launch_bn_bwd_finalize(partial, tbn_bn_bwd_parts(M, c.cout), M, c.cout, scale + c.c_off,
                                         mean + c.c_off, rstd + c.c_off, coef, bn_grad ? g->dgamma + c.c_off : nullptr,
                                         bn_grad ? g->dbeta + c.c_off : nullptr, g->dbias + c.c_off, st));
    if (c.fuse_pool >= 0) {
      const Pool& q = P->pools[c.fuse_pool];
      const Buf& ob = P->bufs[q.outbuf];
      if (!diag_skip(16))
        TBN_TRY(tbn_launch_bn_bwd_apply_pooled(dptr(q.outbuf) + q.out_choff, ob.C,
                                               (const uint8_t*)workspace + q.argmax_off, R, c.outH, c.outW, ob.H, ob.W,
                                               q.stride, q.pad, y, c.cout, scale + c.c_off, shift + c.c_off, coef, y,
                                               st));
    } else if (!diag_skip(16)) {
      TBN_TRY(tbn_launch_bn_bwd_apply(dz, c.nparts, y, M, c.cout, scale + c.c_off, shift + c.c_off, coef, y, st));
    }
    }
    if (c.post_pool && !diag_skip(4))   // the 3x3 average is self-adjoint: d(conv output) = avg_pool(d(pooled))
      TBN_TRY(tbn_launch_avgpool3_fwd(y, c.cout, dconv, c.cout, R, c.outH, c.outW, c.cout, 0, st));
    // weight gradient -- on the aux stream when given: it only reads dy (final after bn_bwd_apply) and
    // the layer input, so it overlaps the data-gradient / BN-backward chain that continues on `st`
    tbn_prof_label(("wgrad " + c.names[c.nparts - 1]).c_str());
    hipStream_t wst = st;
    if (aux != nullptr) {
      hipEvent_t e = PM->ev[ev_next];
      ev_next = (ev_next + 1) & 7;
      (void)hipEventRecord(e, st);
      (void)hipStreamWaitEvent(aux, e, 0);
      wst = aux;
    }
    {
      WgradP wp;
      fill_wgrad(P, c, ws, R, &wp);
      if (c.stem) {
        float* dwp = ws + P->dwpack_off;
        TBN_TRY(tbn_launch_wgrad(wp, 1, dwp, ws + P->wsplit_off, wst));
        if (P->s2d)
          TBN_TRY(tbn_launch_unpack_stem_wgrad_s2d(dwp, g->dweight + c.w_off, 64, wst));
        else
          TBN_TRY(tbn_launch_unpack_stem_wgrad(dwp, g->dweight + c.w_off, 64, P->cin0, P->cp, P->kw, wst));
      } else {
        TBN_TRY(tbn_launch_wgrad(wp, 0, g->dweight + c.w_off, ws + P->wsplit_off, wst));
      }
    }
    // data gradient: conv of dy with flipped / transposed weights (zero-insertion for stride 2)
    if (c.need_dgrad) {
      tbn_prof_label(("dgrad " + c.names[c.nparts - 1]).c_str());
      const float* wt = ws + P->wt_off + c.w_off;
      ConvP p;
      memset(&p, 0, sizeof(p));
      p.in = dconv;
      p.in_ld = c.cout;
      p.wt = wt;
      p.N = R;
      p.H = c.outH;
      p.W = c.outW;
      p.OH = c.inH;
      p.OW = c.inW;
      p.Cin = c.cout;
      p.Cout = c.cin;
      p.R = p.S = c.k;
      p.stride = 1;
      p.pad = c.k - 1 - c.pad;
      p.up = c.stride;
      p.M = R * c.inH * c.inW;
      p.K = c.k * c.k * c.cout;
      p.alg_flops = 2.0 * M * (double)c.cout * c.k * c.k * c.cin;  // = forward count (zero-insertion not counted)
      p.mode = CONV_EPI_PLAIN;
      p.flags = c.dgrad_accum ? CONV_FLAG_ACCUM : 0;
      p.nseg = 1;
      p.seg[0].ptr = ws + ib.doff;
      p.seg[0].ld = ib.C;
      p.seg[0].col_begin = 0;
      p.stages = c.d_stages;
      TBN_TRY(tbn_launch_conv(p, 0, c.d_mt, c.d_nt, st));
    }
  }
  if (aux != nullptr) {  // join: everything the caller enqueues on `st` next sees the weight gradients
    hipEvent_t e = PM->ev[ev_next];
    (void)hipEventRecord(e, aux);
    (void)hipStreamWaitEvent(st, e, 0);
  }
  return TBN_OK;
}

}  // extern "C"
