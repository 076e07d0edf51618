// GRU / LSTM cell backward at one (row, unit) point; shared by the standalone kernel
// (rnn.hip) and the fused epilogue of the backward step product (skinny.hip).
#pragma once
#include "common.hpp"

namespace cvae {

// gp: saved activations of this (row, unit): gp[0], gp[H], gp[2H], gp[3H]
// ox / oh: x-side / h-side pre-activation gradients (oh may be null for LSTM: identical)
// returns the direct (non-GEMM) contribution to dh_{t-1}; LSTM also updates dc (in/out).
template <int CELL>
__device__ __forceinline__ float cell_bwd_point(bool live, float dh, float& dc, const float* __restrict__ gp, int H,
                                                float hprev, float cprev, float ccur, float* __restrict__ ox,
                                                float* __restrict__ oh) {
  if constexpr (CELL == CELL_GRU) {
    if (!live) {
      ox[0] = ox[H] = ox[2 * H] = 0.f;
      oh[0] = oh[H] = oh[2 * H] = 0.f;
      return dh;
    }
    const float rg = gp[0], zg = gp[H], nn = gp[2 * H], hn = gp[3 * H];
    const float dn_pre = dh * (1.f - zg) * (1.f - nn * nn);
    const float dz_pre = dh * (hprev - nn) * zg * (1.f - zg);
    const float dr_pre = dn_pre * hn * rg * (1.f - rg);
    ox[0] = dr_pre; ox[H] = dz_pre; ox[2 * H] = dn_pre;
    oh[0] = dr_pre; oh[H] = dz_pre; oh[2 * H] = dn_pre * rg;
    return dh * zg;
  } else {
    if (!live) {
      ox[0] = ox[H] = ox[2 * H] = ox[3 * H] = 0.f;
      if (oh) oh[0] = oh[H] = oh[2 * H] = oh[3 * H] = 0.f;
      return dh;                       // dc unchanged
    }
    const float ig = gp[0], fg = gp[H], gg = gp[2 * H], og = gp[3 * H];
    const float tc = tanhf_(ccur);
    const float dct = dc + dh * og * (1.f - tc * tc);
    const float di = dct * gg * ig * (1.f - ig);
    const float df = dct * cprev * fg * (1.f - fg);
    const float dg = dct * ig * (1.f - gg * gg);
    const float dout = dh * tc * og * (1.f - og);
    ox[0] = di; ox[H] = df; ox[2 * H] = dg; ox[3 * H] = dout;
    if (oh) { oh[0] = di; oh[H] = df; oh[2 * H] = dg; oh[3 * H] = dout; }
    dc = dct * fg;
    return 0.f;
  }
}

}  // namespace cvae
