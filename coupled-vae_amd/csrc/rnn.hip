// GRU / LSTM sequence drivers (PyTorch gate order and cell equations; reference:
// nn.GRU / nn.LSTM as used by CV/Modules/encoder.py:27-39 and CV/Modules/vae_couple.py:58-70).
//
// Forward, per timestep:
//   R <= SKINNY_MAX_ROWS : ONE launch, skinny MFMA product + fused cell epilogue (skinny.hip)
//   larger R (IW eval)   : tiled MFMA GEMM hp = h_{t-1} W_hh^T, then a bandwidth-bound fused
//                          cell kernel (reads hp, x-side pre-activations, h_{t-1}; writes h_t).
// Backward, per timestep (reverse order): a fused pointwise kernel turns dh_t (+dc_t) and the
// saved gate activations into the pre-activation gradients and the direct carry, then
// dh_{t-1} += dG_h W_hh is an accumulate-GEMM.  Weight gradients are big GEMMs over all
// timesteps at once, done by the caller (api.hip).
// Packed sequences (encoder) are handled by masking: a row is updated only while t < len.
#include "common.hpp"

namespace cvae {

// ------------------------------------------------------------------ forward cell (large R)
template <int CELL>
__global__ __launch_bounds__(256) void cell_fwd_kernel(
    int R, int H, const float* __restrict__ hp, const float* __restrict__ xproj, int xmod,
    const float* __restrict__ xadd, const float* __restrict__ b_hh,
    const float* __restrict__ h_prev, const float* __restrict__ c_prev,
    float* __restrict__ h_out, float* __restrict__ c_out, float* __restrict__ gates,
    const int64_t* __restrict__ lengths, int t) {
  constexpr int G = CELL == CELL_GRU ? 3 : 4;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)R * H) return;
  const int r = (int)(idx / H), n = (int)(idx % H);
  const bool live = (lengths == nullptr) || ((int64_t)t < lengths[r]);
  const float* hpr = hp + (long)r * G * H + n;
  const float* xp = xproj + (long)(r % xmod) * G * H + n;
  const float* xa = xadd ? xadd + (long)r * G * H + n : nullptr;
  float x[G], s[G];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    x[g] = xp[g * H] + (xa ? xa[g * H] : 0.f);
    s[g] = hpr[g * H] + b_hh[g * H + n];
  }
  const float hprev = h_prev[idx];
  if constexpr (CELL == CELL_GRU) {
    const float rg = sigmoidf_(x[0] + s[0]);
    const float zg = sigmoidf_(x[1] + s[1]);
    const float nn = tanhf_(x[2] + rg * s[2]);
    h_out[idx] = live ? (1.f - zg) * nn + zg * hprev : hprev;
    if (gates) {
      float* gp = gates + (long)r * 4 * H + n;
      gp[0] = rg; gp[H] = zg; gp[2 * H] = nn; gp[3 * H] = s[2];
    }
  } else {
    const float ig = sigmoidf_(x[0] + s[0]);
    const float fg = sigmoidf_(x[1] + s[1]);
    const float gg = tanhf_(x[2] + s[2]);
    const float og = sigmoidf_(x[3] + s[3]);
    const float cprev = c_prev[idx];
    const float cnew = fg * cprev + ig * gg;
    h_out[idx] = live ? og * tanhf_(cnew) : hprev;
    c_out[idx] = live ? cnew : cprev;
    if (gates) {
      float* gp = gates + (long)r * 4 * H + n;
      gp[0] = ig; gp[H] = fg; gp[2 * H] = gg; gp[3 * H] = og;
    }
  }
}

int launch_rnn_forward(const RnnFwd& a, hipStream_t s) {
  CVAE_REQUIRE(a.cell == CELL_GRU || a.cell == CELL_LSTM, "rnn_forward: bad cell %d", a.cell);
  CVAE_REQUIRE(a.T >= 0 && a.R > 0 && a.H > 0 && a.xmod > 0, "rnn_forward: bad shape T=%d R=%d H=%d", a.T, a.R, a.H);
  CVAE_REQUIRE(a.xproj && a.W_hh && a.b_hh && a.hs, "rnn_forward: null operand");
  CVAE_REQUIRE(a.cell == CELL_GRU || a.cs, "rnn_forward: LSTM needs cs");
  CVAE_REQUIRE(a.R % a.xmod == 0, "rnn_forward: R must be a multiple of xmod");
  const int G = a.cell == CELL_GRU ? 3 : 4;
  const long RH = (long)a.R * a.H;
  const bool small = a.R <= SKINNY_MAX_ROWS;
  CVAE_REQUIRE(small || a.hp_scratch, "rnn_forward: hp_scratch required for R > %d", SKINNY_MAX_ROWS);
  const int init_slot = a.reverse ? a.T : 0;
  if (a.init_in_place) {
    // nothing to do
  } else if (a.h0) CVAE_TRY(check_hip(hipMemcpyAsync(a.hs + init_slot * RH, a.h0, RH * 4, hipMemcpyDeviceToDevice, s), "h0 copy"));
  else CVAE_TRY(check_hip(hipMemsetAsync(a.hs + init_slot * RH, 0, RH * 4, s), "h0 memset"));
  if (a.cell == CELL_LSTM && !a.init_in_place) {
    if (a.c0) CVAE_TRY(check_hip(hipMemcpyAsync(a.cs + init_slot * RH, a.c0, RH * 4, hipMemcpyDeviceToDevice, s), "c0 copy"));
    else CVAE_TRY(check_hip(hipMemsetAsync(a.cs + init_slot * RH, 0, RH * 4, s), "c0 memset"));
  }
  for (int i = 0; i < a.T; ++i) {
    const int t = a.reverse ? a.T - 1 - i : i;
    const int in_slot = a.reverse ? t + 1 : t, out_slot = a.reverse ? t : t + 1;
    const float* hprev = a.hs + in_slot * RH;
    const float* cprev = a.cs ? a.cs + in_slot * RH : nullptr;
    float* hout = a.hs + out_slot * RH;
    float* cout = a.cs ? a.cs + out_slot * RH : nullptr;
    const float* xp = a.xproj + (long)t * a.xmod * G * a.H;
    float* gates = a.gates ? a.gates + (long)t * a.R * 4 * a.H : nullptr;
    if (small) {
      SkinnyArgs k = {};
      k.A = hprev; k.lda = a.H; k.B = a.W_hh; k.ldb = a.H; k.transB = 1;
      k.M = a.R; k.N = a.H; k.K = a.H; k.ngroups = G; k.gstride = a.H;
      k.xproj = xp; k.ldx = (long)G * a.H; k.xmod = a.xmod; k.xadd = a.xadd; k.b_hh = a.b_hh;
      k.h_prev = hprev; k.c_prev = cprev; k.h_out = hout; k.c_out = cout;
      k.gates = gates; k.ldg = 4L * a.H; k.lengths = a.lengths; k.t = t;
      CVAE_TRY(launch_skinny(k, a.cell == CELL_GRU ? EPI_GRU : EPI_LSTM, s));
    } else {
      GemmDesc d = {};
      d.transA = 0; d.transB = 1; d.M = a.R; d.N = G * a.H; d.K = a.H;
      d.A = hprev; d.lda = a.H; d.B = a.W_hh; d.ldb = a.H; d.C = a.hp_scratch; d.ldc = (long)G * a.H;
      CVAE_TRY(launch_gemm(d, s));
      const int grid = cdiv(RH, 256);
      // algorithmic bytes: hp + x-side (G each) + h_prev in, h out (+ c in/out for LSTM)
      ProfScope prof(PK_CELL_FWD, s, 4.0 * RH * (2 * G + 2 + (a.cell == CELL_LSTM ? 2 : 0)));
      if (a.cell == CELL_GRU)
        hipLaunchKernelGGL((cell_fwd_kernel<CELL_GRU>), dim3(grid), dim3(256), 0, s, a.R, a.H, a.hp_scratch, xp,
                           a.xmod, a.xadd, a.b_hh, hprev, cprev, hout, cout, gates, a.lengths, t);
      else
        hipLaunchKernelGGL((cell_fwd_kernel<CELL_LSTM>), dim3(grid), dim3(256), 0, s, a.R, a.H, a.hp_scratch, xp,
                           a.xmod, a.xadd, a.b_hh, hprev, cprev, hout, cout, gates, a.lengths, t);
      CVAE_LAUNCH_CHECK("cell_fwd_kernel");
    }
  }
  return 0;
}

// ------------------------------------------------------------------ backward cell
// dh_in = dh_acc + dhs_t.  Writes the pre-activation gradients of step t and the direct
// (non-GEMM) part of dh_{t-1} / dc_{t-1} into the *next* accumulators.
template <int CELL>
__global__ __launch_bounds__(256) void cell_bwd_kernel(
    int R, int H, const float* __restrict__ dh_acc, const float* __restrict__ dc_acc,
    const float* __restrict__ dhs_t, const float* __restrict__ gates,
    const float* __restrict__ h_prev, const float* __restrict__ c_prev, const float* __restrict__ c_cur,
    float* __restrict__ dgx, float* __restrict__ dgh, float* __restrict__ dh_next,
    float* __restrict__ dc_next, const int64_t* __restrict__ lengths, int t) {
  constexpr int G = CELL == CELL_GRU ? 3 : 4;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)R * H) return;
  const int r = (int)(idx / H), n = (int)(idx % H);
  const bool live = (lengths == nullptr) || ((int64_t)t < lengths[r]);
  const float dh = dh_acc[idx] + (dhs_t ? dhs_t[idx] : 0.f);
  const float* gp = gates + (long)r * 4 * H + n;
  float* ox = dgx + (long)r * G * H + n;
  float* oh = dgh ? dgh + (long)r * G * H + n : nullptr;
  if constexpr (CELL == CELL_GRU) {
    if (!live) {
      ox[0] = ox[H] = ox[2 * H] = 0.f;
      oh[0] = oh[H] = oh[2 * H] = 0.f;
      dh_next[idx] = dh;
      return;
    }
    const float rg = gp[0], zg = gp[H], nn = gp[2 * H], hn = gp[3 * H];
    const float hprev = h_prev[idx];
    const float dn_pre = dh * (1.f - zg) * (1.f - nn * nn);
    const float dz_pre = dh * (hprev - nn) * zg * (1.f - zg);
    const float dr_pre = dn_pre * hn * rg * (1.f - rg);
    ox[0] = dr_pre; ox[H] = dz_pre; ox[2 * H] = dn_pre;
    oh[0] = dr_pre; oh[H] = dz_pre; oh[2 * H] = dn_pre * rg;
    dh_next[idx] = dh * zg;
  } else {
    const float dc_in = dc_acc[idx];
    if (!live) {
      ox[0] = ox[H] = ox[2 * H] = ox[3 * H] = 0.f;
      if (oh) oh[0] = oh[H] = oh[2 * H] = oh[3 * H] = 0.f;
      dh_next[idx] = dh;
      dc_next[idx] = dc_in;
      return;
    }
    const float ig = gp[0], fg = gp[H], gg = gp[2 * H], og = gp[3 * H];
    const float tc = tanhf_(c_cur[idx]);
    const float dc = dc_in + dh * og * (1.f - tc * tc);
    const float di = dc * gg * ig * (1.f - ig);
    const float df = dc * c_prev[idx] * fg * (1.f - fg);
    const float dg = dc * ig * (1.f - gg * gg);
    const float dout = dh * tc * og * (1.f - og);
    ox[0] = di; ox[H] = df; ox[2 * H] = dg; ox[3 * H] = dout;
    if (oh) { oh[0] = di; oh[H] = df; oh[2 * H] = dg; oh[3 * H] = dout; }
    dh_next[idx] = 0.f;
    dc_next[idx] = dc * fg;
  }
}

int launch_rnn_backward(const RnnBwd& a, hipStream_t s) {
  CVAE_REQUIRE(a.cell == CELL_GRU || a.cell == CELL_LSTM, "rnn_backward: bad cell %d", a.cell);
  CVAE_REQUIRE(a.T >= 0 && a.R > 0 && a.H > 0, "rnn_backward: bad shape");
  CVAE_REQUIRE(a.hs && a.gates && a.W_hh && a.dgx && a.dh0 && a.dh_tmp, "rnn_backward: null operand");
  CVAE_REQUIRE(a.cell == CELL_LSTM || a.dgh, "rnn_backward: GRU needs a separate dgh");
  CVAE_REQUIRE(a.cell == CELL_GRU || (a.cs && a.dc0 && a.dc_tmp), "rnn_backward: LSTM needs cs / dc0 / dc_tmp");
  const int G = a.cell == CELL_GRU ? 3 : 4;
  const long RH = (long)a.R * a.H;
  float* dhb[2] = {a.dh0, a.dh_tmp};
  float* dcb[2] = {a.dc0, a.dc_tmp};
  int cur = a.T & 1;
  if (a.dh_last) CVAE_TRY(check_hip(hipMemcpyAsync(dhb[cur], a.dh_last, RH * 4, hipMemcpyDeviceToDevice, s), "dh_last copy"));
  else CVAE_TRY(check_hip(hipMemsetAsync(dhb[cur], 0, RH * 4, s), "dh memset"));
  if (a.cell == CELL_LSTM) {
    if (a.dc_last) CVAE_TRY(check_hip(hipMemcpyAsync(dcb[cur], a.dc_last, RH * 4, hipMemcpyDeviceToDevice, s), "dc_last copy"));
    else CVAE_TRY(check_hip(hipMemsetAsync(dcb[cur], 0, RH * 4, s), "dc memset"));
  }
  const float* dgh_base = a.dgh ? a.dgh : a.dgx;
  for (int i = a.T - 1; i >= 0; --i) {      // undo the forward order
    const int t = a.reverse ? a.T - 1 - i : i;
    const int in_slot = a.reverse ? t + 1 : t, out_slot = a.reverse ? t : t + 1;
    const long goff = (long)t * a.R * G * a.H;
    const int grid = cdiv(RH, 256);
    const float* dhs_t = a.dhs ? a.dhs + (long)t * RH : nullptr;
    const float* gates = a.gates + (long)t * a.R * 4 * a.H;
    {
    ProfScope prof(PK_CELL_BWD, s, 4.0 * RH * (4 + 2 * G + 4));
    if (a.cell == CELL_GRU)
      hipLaunchKernelGGL((cell_bwd_kernel<CELL_GRU>), dim3(grid), dim3(256), 0, s, a.R, a.H, dhb[cur], nullptr, dhs_t,
                         gates, a.hs + in_slot * RH, nullptr, nullptr, a.dgx + goff, a.dgh + goff, dhb[cur ^ 1],
                         nullptr, a.lengths, t);
    else
      hipLaunchKernelGGL((cell_bwd_kernel<CELL_LSTM>), dim3(grid), dim3(256), 0, s, a.R, a.H, dhb[cur], dcb[cur], dhs_t,
                         gates, a.hs + in_slot * RH, a.cs + in_slot * RH, a.cs + out_slot * RH, a.dgx + goff,
                         a.dgh ? a.dgh + goff : nullptr, dhb[cur ^ 1], dcb[cur ^ 1], a.lengths, t);
    }
    CVAE_LAUNCH_CHECK("cell_bwd_kernel");
    // dh_{t-1} += dG_h[t] . W_hh      (R x GH) . (GH x H)
    if (a.R <= SKINNY_MAX_ROWS) {
      SkinnyArgs k = {};
      k.A = dgh_base + goff; k.lda = (long)G * a.H; k.B = a.W_hh; k.ldb = a.H; k.transB = 0;
      k.M = a.R; k.N = a.H; k.K = G * a.H; k.ngroups = 1; k.gstride = 0;
      k.C = dhb[cur ^ 1]; k.ldc = a.H; k.bias = nullptr; k.act = ACT_NONE; k.beta = 1;
      CVAE_TRY(launch_skinny(k, EPI_LINEAR, s));
    } else {
      GemmDesc d = {};
      d.transA = 0; d.transB = 0; d.M = a.R; d.N = a.H; d.K = G * a.H;
      d.A = dgh_base + goff; d.lda = (long)G * a.H; d.B = a.W_hh; d.ldb = a.H;
      d.C = dhb[cur ^ 1]; d.ldc = a.H; d.beta = 1;
      CVAE_TRY(launch_gemm(d, s));
    }
    cur ^= 1;
  }
  return 0;
}

}  // namespace cvae
