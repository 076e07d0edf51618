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
#include "misc.hpp"
#include "cell_math.hpp"

namespace cvae {

// ------------------------------------------------------------------ forward cell (large R)
template <int CELL>
__global__ __launch_bounds__(256) void cell_fwd_kernel(
    int R, int H, const float* __restrict__ hp, const float* __restrict__ xproj, int xmod, int xdiv,
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
  const float* xp = xproj + (long)((r / xdiv) % xmod) * G * H + n;
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

static int rnn_forward_check(const RnnFwd& a) {
  CVAE_REQUIRE(a.cell == CELL_GRU || a.cell == CELL_LSTM, "rnn_forward: bad cell %d", a.cell);
  CVAE_REQUIRE(a.T >= 0 && a.R > 0 && a.H > 0 && a.xmod > 0, "rnn_forward: bad shape T=%d R=%d H=%d", a.T, a.R, a.H);
  CVAE_REQUIRE(a.xproj && a.W_hh && a.b_hh && a.hs, "rnn_forward: null operand");
  CVAE_REQUIRE(a.cell == CELL_GRU || a.cs, "rnn_forward: LSTM needs cs");
  CVAE_REQUIRE(a.R % a.xmod == 0, "rnn_forward: R must be a multiple of xmod");
  CVAE_REQUIRE((a.xdiv <= 1 && !a.live_rows) || a.R > SKINNY_MAX_ROWS,
               "rnn_forward: xdiv / live_rows need R > %d", SKINNY_MAX_ROWS);
  CVAE_REQUIRE(a.xdiv <= 1 || a.R % ((long)a.xdiv * a.xmod) == 0, "rnn_forward: R must be a multiple of xdiv*xmod");
  CVAE_REQUIRE(a.R <= SKINNY_MAX_ROWS || a.hp_scratch, "rnn_forward: hp_scratch required for R > %d", SKINNY_MAX_ROWS);
  return 0;
}

// zero initial states are queued on `zb` (flushed by the caller: one launch for all sequences)
static int rnn_forward_init(const RnnFwd& a, hipStream_t s, ZeroBatch& zb) {
  const long RH = (long)a.R * a.H;
  const int init_slot = a.reverse ? a.T : 0;
  if (a.init_in_place) return 0;
  if (a.h0) CVAE_TRY(launch_copy(a.hs + init_slot * RH, a.h0, RH, s));
  else zb.add(a.hs + init_slot * RH, RH);
  if (a.cell == CELL_LSTM) {
    if (a.c0) CVAE_TRY(launch_copy(a.cs + init_slot * RH, a.c0, RH, s));
    else zb.add(a.cs + init_slot * RH, RH);
  }
  return 0;
}

// arguments of step i (processing order) of one sequence
static void rnn_forward_step_args(const RnnFwd& a, int i, SkinnyArgs& k) {
  const int G = a.cell == CELL_GRU ? 3 : 4;
  const long RH = (long)a.R * a.H;
  const int t = a.reverse ? a.T - 1 - i : i;
  const int in_slot = a.reverse ? t + 1 : t, out_slot = a.reverse ? t : t + 1;
  k = SkinnyArgs{};
  k.A = a.hs + in_slot * RH; k.lda = a.H; k.B = a.W_hh; k.ldb = a.H; k.transB = 1;
  k.M = a.R; k.N = a.H; k.K = a.H; k.ngroups = G; k.gstride = a.H;
  k.xproj = a.xproj + (long)t * a.xmod * G * a.H; k.ldx = (long)G * a.H; k.xmod = a.xmod; k.xadd = a.xadd;
  k.b_hh = a.b_hh;
  k.h_prev = k.A; k.c_prev = a.cs ? a.cs + in_slot * RH : nullptr;
  k.h_out = a.hs + out_slot * RH; k.c_out = a.cs ? a.cs + out_slot * RH : nullptr;
  k.gates = a.gates ? a.gates + (long)t * a.R * 4 * a.H : nullptr; k.ldg = 4L * a.H;
  k.lengths = a.lengths; k.t = t;
}

static int rnn_forward_large(const RnnFwd& a, hipStream_t s) {
  const int G = a.cell == CELL_GRU ? 3 : 4;
  const int xdiv = a.xdiv > 1 ? a.xdiv : 1;
  // bf16 (hi, lo) images of h for the weights-in-registers step kernel: two (step parity), library scratch
  char* img[2] = {nullptr, nullptr};
  void* wimg = nullptr;                   // W_hh as that kernel's register fragments (written once per call)
  const long img_lo = rnn_large_image_lo(a.R, a.H);
  bool img_valid = false;                 // img[i & 1] holds the split of step i's h_prev ...
  int img_rows = 0;                       // ... for its first img_rows rows
  for (int i = 0; i < a.T; ++i) {
    SkinnyArgs k;
    rnn_forward_step_args(a, i, k);
    const int live = a.live_rows ? a.live_rows[k.t] : a.R;
    CVAE_REQUIRE(live >= 0 && live <= a.R, "rnn_forward: live_rows[%d]=%d outside [0,%d]", k.t, live, a.R);
    if (live == 0) continue;
    if (!a.gates && !a.lengths && rnn_large_wreg_supported(a.cell, a.H, live, a.xmod, xdiv, k.A, a.W_hh)) {
      if (!img[0]) {
        void* base = nullptr;
        const size_t one = rnn_large_image_bytes(a.R, a.H);
        CVAE_TRY(lib_scratch(LS_H_IMAGE, s, 2 * one, &base));
        img[0] = static_cast<char*>(base);
        img[1] = img[0] + one;
        CVAE_TRY(lib_scratch(LS_W_IMAGE, s, rnn_large_wimg_bytes(a.H), &wimg));
        CVAE_TRY(launch_rnn_large_wimg(a.cell, a.W_hh, a.H, wimg, s));
      }
      // (a live prefix that grows again reads rows the previous step's cell did not write: split those from h_prev)
      if (!img_valid || live > img_rows) CVAE_TRY(launch_rnn_large_split(k.h_prev, live, a.H, img[i & 1], img_lo, s));
      img_rows = live;
      CVAE_TRY(launch_rnn_large_wreg_step(a.cell, live, a.H, k.h_prev, k.c_prev, wimg, a.b_hh, k.xproj, a.xmod, xdiv, a.xadd,
                                          k.h_out, k.c_out, img[i & 1], img[(i + 1) & 1], img_lo, s));
      img_valid = true;
      continue;
    }
    img_valid = false;
    if (!a.gates && !a.lengths && rnn_large_fused_supported(a.cell, a.H, k.A, a.W_hh)) {
      CVAE_TRY(launch_rnn_large_step(a.cell, live, a.H, k.h_prev, k.c_prev, a.W_hh, a.b_hh, k.xproj, a.xmod, xdiv,
                                     a.xadd, k.h_out, k.c_out, s));
      continue;
    }
    const long RH = (long)live * a.H;
    GemmDesc d = {};
    d.transA = 0; d.transB = 1; d.M = live; d.N = G * a.H; d.K = a.H;
    d.A = k.A; d.lda = a.H; d.B = a.W_hh; d.ldb = a.H; d.C = a.hp_scratch; d.ldc = (long)G * a.H;
    CVAE_TRY(launch_gemm(d, s));
    const int grid = cdiv(RH, 256);
    // algorithmic bytes: hp + x-side (G each) + h_prev in, h out (+ c in/out for LSTM)
    ProfScope prof(PK_CELL_FWD, s, 4.0 * RH * (2 * G + 2 + (a.cell == CELL_LSTM ? 2 : 0)));
    if (a.cell == CELL_GRU)
      hipLaunchKernelGGL((cell_fwd_kernel<CELL_GRU>), dim3(grid), dim3(256), 0, s, live, a.H, a.hp_scratch, k.xproj,
                         a.xmod, xdiv, a.xadd, a.b_hh, k.h_prev, k.c_prev, k.h_out, k.c_out, k.gates, a.lengths, k.t);
    else
      hipLaunchKernelGGL((cell_fwd_kernel<CELL_LSTM>), dim3(grid), dim3(256), 0, s, live, a.H, a.hp_scratch, k.xproj,
                         a.xmod, xdiv, a.xadd, a.b_hh, k.h_prev, k.c_prev, k.h_out, k.c_out, k.gates, a.lengths, k.t);
    CVAE_LAUNCH_CHECK("cell_fwd_kernel");
  }
  return 0;
}

// Weights far beyond L2 (stream_rnn_supported): one launch per step of the weight-streaming split-bf16 kernel,
// W_hh of every sequence pre-split once into its fragment-ordered bf16 hi / lo image (library scratch).
static int rnn_forward_stream(const RnnFwd* a, int ns, hipStream_t s) {
  const int G = a[0].cell == CELL_GRU ? 3 : 4, H = a[0].H;
  const size_t img_bytes = stream_image_bytes(G * H, H);
  void* base = nullptr;
  CVAE_TRY(lib_scratch(LS_STREAM_IMG, s, ns * img_bytes, &base));
  void* imgs[SKINNY_MAX_BATCH];
  for (int q = 0; q < ns; ++q) {
    imgs[q] = static_cast<char*>(base) + q * img_bytes;
    CVAE_TRY(launch_stream_presplit(a[q].W_hh, H, 1, G * H, H, G, 16, H, imgs[q], s));
  }
  SkinnyArgs k[SKINNY_MAX_BATCH];
  for (int i = 0; i < a[0].T; ++i) {
    for (int q = 0; q < ns; ++q) rnn_forward_step_args(a[q], i, k[q]);
    CVAE_TRY(launch_skinny_stream(k, imgs, ns, a[0].cell == CELL_GRU ? EPI_GRU : EPI_LSTM, 1, nullptr, s));
  }
  return 0;
}

int launch_rnn_forward_multi(const RnnFwd* a, int ns, hipStream_t s) {
  CVAE_REQUIRE(ns >= 1 && ns <= SKINNY_MAX_BATCH, "rnn_forward: %d sequences (max %d)", ns, SKINNY_MAX_BATCH);
  ZeroBatch zb(s);
  for (int q = 0; q < ns; ++q) {
    CVAE_TRY(rnn_forward_check(a[q]));
    CVAE_REQUIRE(a[q].cell == a[0].cell && a[q].T == a[0].T && a[q].R == a[0].R && a[q].H == a[0].H,
                 "rnn_forward: lock-step sequences differ in shape");
    CVAE_TRY(rnn_forward_init(a[q], s, zb));
  }
  if (a[0].R > SKINNY_MAX_ROWS) {
    CVAE_TRY(zb.flush());
    for (int q = 0; q < ns; ++q) CVAE_TRY(rnn_forward_large(a[q], s));
    return 0;
  }
  bool aligned = true;                    // the whole-sequence kernel reads and writes with 16-byte accesses
  for (int q = 0; q < ns; ++q)
    aligned = aligned && aligned16(a[q].hs) && (!a[q].cs || aligned16(a[q].cs)) &&
              (!a[q].gates || aligned16(a[q].gates)) && aligned16(a[q].W_hh);
  if (aligned && rnn_persist_supported(a[0].cell, a[0].T, a[0].R, a[0].H, ns))
    return launch_rnn_forward_persist(a, ns, s, &zb);      // its flag words join the same zero launch
  CVAE_TRY(zb.flush());
  // the lock-stepped sequences together need more CUs than there are free ones (cvae_persist_headroom: a
  // data-parallel job keeps some for RCCL's kernels), but half of them fit: two whole-sequence launches
  if (aligned && ns > 1 && rnn_persist_supported(a[0].cell, a[0].T, a[0].R, a[0].H, (ns + 1) / 2)) {
    const int h = (ns + 1) / 2;
    CVAE_TRY(launch_rnn_forward_persist(a, h, s, nullptr));
    return launch_rnn_forward_persist(a + h, ns - h, s, nullptr);
  }
  if (aligned && stream_rnn_supported(a[0].R, a[0].H, a[0].cell == CELL_GRU ? 3 : 4)) return rnn_forward_stream(a, ns, s);
  SkinnyArgs k[SKINNY_MAX_BATCH];
  for (int i = 0; i < a[0].T; ++i) {
    for (int q = 0; q < ns; ++q) rnn_forward_step_args(a[q], i, k[q]);
    CVAE_TRY(launch_skinny_multi(k, ns, a[0].cell == CELL_GRU ? EPI_GRU : EPI_LSTM, s));
  }
  return 0;
}

int launch_rnn_forward(const RnnFwd& a, hipStream_t s) { return launch_rnn_forward_multi(&a, 1, s); }

// ------------------------------------------------------------------ backward cell
// dh_in = dh_acc + dhs_t.  Writes the pre-activation gradients of step t and the direct
// (non-GEMM) part of dh_{t-1} / dc_{t-1} into the *next* accumulators (may alias the inputs).
template <int CELL>
__global__ __launch_bounds__(256) void cell_bwd_kernel(
    int R, int H, const float* dh_acc, const float* dc_acc, const float* __restrict__ dhs_t,
    const float* __restrict__ gates, const float* __restrict__ h_prev, const float* __restrict__ c_prev,
    const float* __restrict__ c_cur, float* __restrict__ dgx, float* __restrict__ dgh, float* dh_next,
    float* dc_next, const int64_t* __restrict__ lengths, int t) {
  constexpr int G = CELL == CELL_GRU ? 3 : 4;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)R * H) return;
  const int r = (int)(idx / H), n = (int)(idx % H);
  const bool live = (lengths == nullptr) || ((int64_t)t < lengths[r]);
  const float dh = dh_acc[idx] + (dhs_t ? dhs_t[idx] : 0.f);
  float dc = (CELL == CELL_LSTM) ? dc_acc[idx] : 0.f;
  const float direct = cell_bwd_point<CELL>(live, dh, dc, gates + (long)r * 4 * H + n, H, h_prev[idx],
                                            CELL == CELL_LSTM ? c_prev[idx] : 0.f,
                                            CELL == CELL_LSTM ? c_cur[idx] : 0.f, dgx + (long)r * G * H + n,
                                            dgh ? dgh + (long)r * G * H + n : nullptr);
  dh_next[idx] = direct;
  if constexpr (CELL == CELL_LSTM) dc_next[idx] = dc;
}

// dst[c, r] = src[r, c]
__global__ __launch_bounds__(256) void transpose_kernel(int R, int C, const float* __restrict__ src,
                                                        float* __restrict__ dst) {
  __shared__ float tile[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int i = ty; i < 32; i += 8)
    if (r0 + i < R && c0 + tx < C) tile[i][tx] = src[(long)(r0 + i) * C + c0 + tx];
  __syncthreads();
  for (int i = ty; i < 32; i += 8)
    if (c0 + i < C && r0 + tx < R) dst[(long)(c0 + i) * R + r0 + tx] = tile[tx][i];
}

static int launch_cell_bwd(const RnnBwd& a, int t, const float* dh_in, const float* dc_in, float* dh_out,
                           float* dc_out, hipStream_t s) {
  const int G = a.cell == CELL_GRU ? 3 : 4;
  const long RH = (long)a.R * a.H;
  const int in_slot = a.reverse ? t + 1 : t, out_slot = a.reverse ? t : t + 1;
  const long goff = (long)t * a.R * G * a.H;
  const int grid = cdiv(RH, 256);
  const float* dhs_t = a.dhs ? a.dhs + (long)t * RH : nullptr;
  const float* gates = a.gates + (long)t * a.R * 4 * a.H;
  ProfScope prof(PK_CELL_BWD, s, 4.0 * RH * (4 + 2 * G + 4));
  if (a.cell == CELL_GRU)
    hipLaunchKernelGGL((cell_bwd_kernel<CELL_GRU>), dim3(grid), dim3(256), 0, s, a.R, a.H, dh_in, nullptr, dhs_t, gates,
                       a.hs + in_slot * RH, nullptr, nullptr, a.dgx + goff, a.dgh + goff, dh_out, nullptr, a.lengths,
                       t);
  else
    hipLaunchKernelGGL((cell_bwd_kernel<CELL_LSTM>), dim3(grid), dim3(256), 0, s, a.R, a.H, dh_in, dc_in, dhs_t, gates,
                       a.hs + in_slot * RH, a.cs + in_slot * RH, a.cs + out_slot * RH, a.dgx + goff,
                       a.dgh ? a.dgh + goff : nullptr, dh_out, dc_out, a.lengths, t);
  CVAE_LAUNCH_CHECK("cell_bwd_kernel");
  return 0;
}

static int rnn_backward_check(const RnnBwd& a) {
  CVAE_REQUIRE(a.cell == CELL_GRU || a.cell == CELL_LSTM, "rnn_backward: bad cell %d", a.cell);
  CVAE_REQUIRE(a.T >= 0 && a.R > 0 && a.H > 0, "rnn_backward: bad shape");
  CVAE_REQUIRE(a.hs && a.gates && a.W_hh && a.dgx && a.dh0, "rnn_backward: null operand");
  CVAE_REQUIRE(a.cell == CELL_LSTM || a.dgh, "rnn_backward: GRU needs a separate dgh");
  CVAE_REQUIRE(a.cell == CELL_GRU || (a.cs && a.dc0), "rnn_backward: LSTM needs cs / dc0");
  return 0;
}

static int rnn_backward_large(const RnnBwd& a, hipStream_t s);

// cell backward of one step for up to four lock-stepped sequences (blockIdx.y); the incoming gradient is
// dh + the K-range partial products of the previous launch (parts[q][p][R*H], p < nparts) + dhs_t.
// gates == nullptr: only folds the partial products into dh (the gradient wrt the initial state).
struct CellBwdSeq {
  const float *dhs_t, *gates, *h_prev, *c_prev, *c_cur;
  float *dgx, *dgh, *dh, *dc;
  const int64_t* lengths;
  int t;
};
struct CellBwdBatch { CellBwdSeq q[SKINNY_MAX_BATCH]; };

template <int CELL>
__global__ __launch_bounds__(256) void cell_bwd_parts_kernel(CellBwdBatch b, int R, int H, const float* __restrict__ parts,
                                                             int nparts) {
  constexpr int G = CELL == CELL_GRU ? 3 : 4;
  const CellBwdSeq& q = b.q[blockIdx.y];
  const long RH = (long)R * H;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= RH) return;
  const int r = (int)(idx / H), n = (int)(idx % H);
  float dh = q.dh[idx];
  if (parts) {
    const float* pp = parts + (long)blockIdx.y * nparts * RH + idx;
    for (int p = 0; p < nparts; ++p) dh += pp[p * RH];
  }
  if (!q.gates) { q.dh[idx] = dh; return; }
  if (q.dhs_t) dh += q.dhs_t[idx];
  const bool live = (q.lengths == nullptr) || ((int64_t)q.t < q.lengths[r]);
  float dc = (CELL == CELL_LSTM) ? q.dc[idx] : 0.f;
  const float direct = cell_bwd_point<CELL>(live, dh, dc, q.gates + (long)r * 4 * H + n, H, q.h_prev[idx],
                                            CELL == CELL_LSTM ? q.c_prev[idx] : 0.f,
                                            CELL == CELL_LSTM ? q.c_cur[idx] : 0.f, q.dgx + (long)r * G * H + n,
                                            q.dgh ? q.dgh + (long)r * G * H + n : nullptr);
  q.dh[idx] = direct;
  if constexpr (CELL == CELL_LSTM) q.dc[idx] = dc;
}

// Weights far beyond L2: per step one cell-backward launch (all sequences) and one weight-streaming product
// dG_t . W_hh split over K ranges (partial products in library scratch, summed by the next cell launch: fixed order).
static int rnn_backward_stream(const RnnBwd* arr, int ns, hipStream_t s) {
  const int T = arr[0].T, H = arr[0].H, R = arr[0].R, cell = arr[0].cell;
  const int G = cell == CELL_GRU ? 3 : 4;
  const long RH = (long)R * H;
  for (int q = 0; q < ns; ++q) {
    const RnnBwd& a = arr[q];
    if (a.dh_last) CVAE_TRY(launch_copy(a.dh0, a.dh_last, RH, s));
    else CVAE_TRY(launch_zero(a.dh0, RH, s));
    if (cell == CELL_LSTM) {
      if (a.dc_last) CVAE_TRY(launch_copy(a.dc0, a.dc_last, RH, s));
      else CVAE_TRY(launch_zero(a.dc0, RH, s));
    }
  }
  if (T == 0) return 0;
  const size_t img_bytes = stream_image_bytes(H, G * H);
  void* base = nullptr;
  CVAE_TRY(lib_scratch(LS_STREAM_IMG, s, ns * img_bytes, &base));
  void* imgs[SKINNY_MAX_BATCH];
  for (int q = 0; q < ns; ++q) {
    imgs[q] = static_cast<char*>(base) + q * img_bytes;
    // op(B)[k = gate row][n = unit] = W_hh[k*H + n]; four adjacent 16-column tiles per block
    CVAE_TRY(launch_stream_presplit(arr[q].W_hh, 1, H, H, G * H, 4, 64, 16, imgs[q], s));
  }
  const int ks = stream_linear_ksplit(ns, H, G * H);
  void* pbuf = nullptr;
  const size_t parts_bytes = ((size_t)ns * ks * RH * sizeof(float) + 15) / 16 * 16;
  CVAE_TRY(lib_scratch(LS_STREAM_PARTS, s, parts_bytes + (size_t)ns * 64 * sizeof(int64_t), &pbuf));
  float* parts = static_cast<float*>(pbuf);
  // unpacked sequences (decoders) that come without a row bound: rows past their last non-zero output gradient
  // stay exactly zero for the rest of the backward pass (as launch_rnn_backward_persist does)
  const int64_t* za[SKINNY_MAX_BATCH];
  for (int q = 0; q < ns; ++q) {
    const RnnBwd& a = arr[q];
    za[q] = a.zero_after ? a.zero_after : a.lengths;
    if (!za[q] && a.dhs && !a.dh_last && !a.dc_last) {
      int64_t* zbuf = reinterpret_cast<int64_t*>(static_cast<char*>(pbuf) + parts_bytes) + (size_t)q * 64;
      CVAE_TRY(launch_last_nonzero_step(T, R, H, a.dhs, zbuf, s));
      za[q] = zbuf;
    }
  }
  const dim3 cgrid(cdiv(RH, 256), ns);
  SkinnyArgs kk[SKINNY_MAX_BATCH];
  for (int i = T - 1; i >= -1; --i) {
    CellBwdBatch cb = {};
    for (int q = 0; q < ns; ++q) {
      const RnnBwd& a = arr[q];
      CellBwdSeq& c = cb.q[q];
      c.dh = a.dh0; c.dc = a.dc0;
      if (i < 0) continue;                                   // last launch: fold the partial products into dh0
      const int t = a.reverse ? T - 1 - i : i;
      const int in_slot = a.reverse ? t + 1 : t, out_slot = a.reverse ? t : t + 1;
      const long goff = (long)t * R * G * H;
      c.dhs_t = a.dhs ? a.dhs + (long)t * RH : nullptr;
      c.gates = a.gates + (long)t * R * 4 * H;
      c.h_prev = a.hs + in_slot * RH;
      c.c_prev = a.cs ? a.cs + in_slot * RH : nullptr;
      c.c_cur = a.cs ? a.cs + out_slot * RH : nullptr;
      c.dgx = a.dgx + goff; c.dgh = a.dgh ? a.dgh + goff : nullptr;
      c.lengths = a.lengths; c.t = t;
      SkinnyArgs& k = kk[q];
      k = SkinnyArgs{};
      k.A = (a.dgh ? a.dgh : a.dgx) + goff; k.lda = (long)G * H;
      k.M = R; k.N = H; k.K = G * H; k.ngroups = 4;
      // rows whose dG_t is exactly zero (a padded step of a packed sequence, a decoder row past its last target)
      // are left out of the product: their partial rows are written as zeros
      k.lengths = za[q]; k.t = t;
    }
    const float* pin = (i == T - 1) ? nullptr : parts;
    {
      ProfScope prof(PK_CELL_BWD, s, 4.0 * RH * ns * (4 + 2 * G + 4 + ks));
      if (cell == CELL_GRU)
        hipLaunchKernelGGL((cell_bwd_parts_kernel<CELL_GRU>), cgrid, dim3(256), 0, s, cb, R, H, pin, ks);
      else
        hipLaunchKernelGGL((cell_bwd_parts_kernel<CELL_LSTM>), cgrid, dim3(256), 0, s, cb, R, H, pin, ks);
      CVAE_LAUNCH_CHECK("cell_bwd_parts_kernel");
    }
    if (i >= 0) CVAE_TRY(launch_skinny_stream(kk, imgs, ns, EPI_LINEAR, ks, parts, s));
  }
  return 0;
}

int launch_rnn_backward_multi(const RnnBwd* arr, int ns, hipStream_t s) {
  CVAE_REQUIRE(ns >= 1 && ns <= SKINNY_MAX_BATCH, "rnn_backward: %d sequences (max %d)", ns, SKINNY_MAX_BATCH);
  for (int q = 0; q < ns; ++q) {
    CVAE_TRY(rnn_backward_check(arr[q]));
    CVAE_REQUIRE(arr[q].cell == arr[0].cell && arr[q].T == arr[0].T && arr[q].R == arr[0].R && arr[q].H == arr[0].H,
                 "rnn_backward: lock-step sequences differ in shape");
  }
  if (arr[0].R > SKINNY_MAX_ROWS) {
    for (int q = 0; q < ns; ++q) CVAE_TRY(rnn_backward_large(arr[q], s));
    return 0;
  }
  bool aligned = true;
  for (int q = 0; q < ns; ++q)
    aligned = aligned && aligned16(arr[q].dgx) && (!arr[q].dgh || aligned16(arr[q].dgh)) && aligned16(arr[q].W_hh);
  if (aligned && rnn_persist_bwd_supported(arr[0].cell, arr[0].T, arr[0].R, arr[0].H, ns))
    return launch_rnn_backward_persist(arr, ns, s);
  if (aligned && ns > 1 && rnn_persist_bwd_supported(arr[0].cell, arr[0].T, arr[0].R, arr[0].H, (ns + 1) / 2)) {
    const int h = (ns + 1) / 2;                            // see launch_rnn_forward_multi
    CVAE_TRY(launch_rnn_backward_persist(arr, h, s));
    return launch_rnn_backward_persist(arr + h, ns - h, s);
  }
  if (aligned && stream_rnn_supported(arr[0].R, arr[0].H, arr[0].cell == CELL_GRU ? 3 : 4)) return rnn_backward_stream(arr, ns, s);
  // ---- one launch per step (for all sequences): dh_{t-1} product with the previous step's cell backward fused in
  const int T = arr[0].T, H = arr[0].H, R = arr[0].R;
  const int G = arr[0].cell == CELL_GRU ? 3 : 4;
  const long RH = (long)R * H;
  for (int q = 0; q < ns; ++q) {
    const RnnBwd& a = arr[q];
    CVAE_REQUIRE(a.whhT, "rnn_backward: whhT scratch required for R <= %d", SKINNY_MAX_ROWS);
    if (a.dh_last) CVAE_TRY(launch_copy(a.dh0, a.dh_last, RH, s));
    else CVAE_TRY(launch_zero(a.dh0, RH, s));
    if (a.cell == CELL_LSTM) {
      if (a.dc_last) CVAE_TRY(launch_copy(a.dc0, a.dc_last, RH, s));
      else CVAE_TRY(launch_zero(a.dc0, RH, s));
    }
    if (T == 0) continue;
    hipLaunchKernelGGL(transpose_kernel, dim3(cdiv(H, 32), cdiv(G * H, 32)), dim3(256), 0, s, G * H, H, a.W_hh, a.whhT);
    CVAE_LAUNCH_CHECK("transpose_kernel");
    CVAE_TRY(launch_cell_bwd(a, a.reverse ? 0 : T - 1, a.dh0, a.dc0, a.dh0, a.dc0, s));
  }
  SkinnyArgs kk[SKINNY_MAX_BATCH];
  for (int i = T - 1; i >= 0; --i) {
    for (int q = 0; q < ns; ++q) {
      const RnnBwd& a = arr[q];
      const float* dgh_base = a.dgh ? a.dgh : a.dgx;
      const int t = a.reverse ? T - 1 - i : i;
      const bool has_prev = i > 0;
      const int tp = a.reverse ? t + 1 : t - 1;          // previous step in forward order
      SkinnyArgs& k = kk[q];
      k = SkinnyArgs{};
      k.A = dgh_base + (long)t * R * G * H; k.lda = (long)G * H;
      k.B = a.whhT; k.ldb = (long)G * H; k.transB = 1;
      k.M = R; k.N = H; k.K = G * H; k.ngroups = 1; k.gstride = 0;
      k.dh_buf = a.dh0; k.dc_buf = a.dc0; k.has_prev = has_prev; k.lengths = a.lengths;
      if (has_prev) {
        const int in_slot = a.reverse ? tp + 1 : tp, out_slot = a.reverse ? tp : tp + 1;
        k.t_prev = tp;
        k.dhs_prev = a.dhs ? a.dhs + (long)tp * RH : nullptr;
        k.gates_prev = a.gates + (long)tp * R * 4 * H;
        k.hin_prev = a.hs + in_slot * RH;
        k.cin_prev = a.cs ? a.cs + in_slot * RH : nullptr;
        k.cout_prev = a.cs ? a.cs + out_slot * RH : nullptr;
        k.dgx_prev = a.dgx + (long)tp * R * G * H;
        k.dgh_prev = a.dgh ? a.dgh + (long)tp * R * G * H : nullptr;
      }
    }
    CVAE_TRY(launch_skinny_multi(kk, ns, arr[0].cell == CELL_GRU ? EPI_GRU_BWD : EPI_LSTM_BWD, s));
  }
  return 0;
}

int launch_rnn_backward(const RnnBwd& a, hipStream_t s) { return launch_rnn_backward_multi(&a, 1, s); }

static int rnn_backward_large(const RnnBwd& a, hipStream_t s) {
  const int G = a.cell == CELL_GRU ? 3 : 4;
  const long RH = (long)a.R * a.H;
  const float* dgh_base = a.dgh ? a.dgh : a.dgx;
  // ---- large R: pointwise cell backward + accumulate-GEMM per step, ping-pong accumulators
  CVAE_REQUIRE(a.dh_tmp && (a.cell == CELL_GRU || a.dc_tmp), "rnn_backward: dh_tmp / dc_tmp scratch required");
  float* dhb[2] = {a.dh0, a.dh_tmp};
  float* dcb[2] = {a.dc0, a.dc_tmp};
  int cur = a.T & 1;
  if (a.dh_last) CVAE_TRY(launch_copy(dhb[cur], a.dh_last, RH, s));
  else CVAE_TRY(launch_zero(dhb[cur], RH, s));
  if (a.cell == CELL_LSTM) {
    if (a.dc_last) CVAE_TRY(launch_copy(dcb[cur], a.dc_last, RH, s));
    else CVAE_TRY(launch_zero(dcb[cur], RH, s));
  }
  for (int i = a.T - 1; i >= 0; --i) {      // undo the forward order
    const int t = a.reverse ? a.T - 1 - i : i;
    const long goff = (long)t * a.R * G * a.H;
    CVAE_TRY(launch_cell_bwd(a, t, dhb[cur], dcb[cur], dhb[cur ^ 1], dcb[cur ^ 1], s));
    GemmDesc d = {};
    d.transA = 0; d.transB = 0; d.M = a.R; d.N = a.H; d.K = G * a.H;
    d.A = dgh_base + goff; d.lda = (long)G * a.H; d.B = a.W_hh; d.ldb = a.H;
    d.C = dhb[cur ^ 1]; d.ldc = a.H; d.beta = 1;
    CVAE_TRY(launch_gemm(d, s));
    cur ^= 1;
  }
  return 0;
}

}  // namespace cvae
