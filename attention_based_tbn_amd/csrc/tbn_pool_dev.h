// 2x2-input-block form of the 3x3 / stride 2 / pad 0 max-pool gradient gather (shared by bn.hip and pool.hip).
#pragma once
#include "tbn_common.h"

struct PoolBlk {
  const float* dout;
  const unsigned char* argmax;
  int dout_ld, H, W, OH, OW, C, BH, BW;
  FastDiv div_bw, div_bh;
};
static inline PoolBlk make_poolblk(const float* dout, int dout_ld, const unsigned char* argmax, int H, int W, int OH, int OW, int C) {
  PoolBlk q;
  q.dout = dout; q.argmax = argmax; q.dout_ld = dout_ld;
  q.H = H; q.W = W; q.OH = OH; q.OW = OW; q.C = C;
  q.BH = (H + 1) / 2; q.BW = (W + 1) / 2;
  q.div_bw = make_fastdiv((uint32_t)q.BW);
  q.div_bh = make_fastdiv((uint32_t)q.BH);
  return q;
}
// g[0] = d z(2by, 2bx), g[1] = d z(2by, 2bx+1), g[2] = d z(2by+1, 2bx), g[3] = d z(2by+1, 2bx+1)
__device__ __forceinline__ void pooled_grad_2x2(const PoolBlk& q, int n, int by, int bx, int c, float4 (&g)[4]) {
  const bool vy[2] = {by < q.OH, by >= 1 && by - 1 < q.OH}, vx[2] = {bx < q.OW, bx >= 1 && bx - 1 < q.OW};
  const int oy[2] = {vy[0] ? by : 0, vy[1] ? by - 1 : 0}, ox[2] = {vx[0] ? bx : 0, vx[1] ? bx - 1 : 0};
  uint32_t am[4];
  float4 d[4];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const size_t opix = (size_t)(n * q.OH + oy[a]) * q.OW + ox[b];
      am[a * 2 + b] = *reinterpret_cast<const uint32_t*>(q.argmax + opix * q.C + c);
      d[a * 2 + b] = *reinterpret_cast<const float4*>(q.dout + opix * q.dout_ld + c);
      if (!(vy[a] && vx[b])) am[a * 2 + b] = 0xffffffffu;   // matches no tap
    }
  auto take = [&](float4& acc, int w, uint32_t k) {
    const uint32_t m = am[w];
    if ((m & 0xff) == k) acc.x += d[w].x;
    if (((m >> 8) & 0xff) == k) acc.y += d[w].y;
    if (((m >> 16) & 0xff) == k) acc.z += d[w].z;
    if ((m >> 24) == k) acc.w += d[w].w;
  };
#pragma unroll
  for (int i = 0; i < 4; ++i) g[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  // window index w = a*2 + b: a = 0 window row by, 1 row by-1; b = 0 window column bx, 1 column bx-1
  take(g[0], 0, 0); take(g[0], 1, 2); take(g[0], 2, 6); take(g[0], 3, 8);   // even-even pixel: taps (0,0) (0,2) (2,0) (2,2)
  take(g[1], 0, 1); take(g[1], 2, 7);                                       // even row, odd column: (0,1) (2,1)
  take(g[2], 0, 3); take(g[2], 1, 5);                                       // odd row, even column: (1,0) (1,2)
  take(g[3], 0, 4);                                                         // odd-odd: (1,1)
}

