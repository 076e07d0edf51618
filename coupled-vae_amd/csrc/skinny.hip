// Few-row (M <= 64 per block) fp32 products on v_mfma_f32_16x16x4_f32.
//
// One block owns 16 output columns (times `ngroups` gate groups) and 16*MT rows.  Its NW
// waves split the K dimension: wave w takes the 16-deep K blocks w, w+NW, ...  Operands go
// straight from L2 to registers (they are read once per block; an LDS round trip would be
// pure overhead, cdna_hip_programming.md "GEMV / M <= 16" row).  The loop is latency-bound
// (a wave has only a handful of K blocks), so loads are issued U K-blocks ahead of the
// MFMAs that consume them, and the epilogue's operands are fetched before the main loop.
// Partial accumulators are summed through LDS and the epilogue runs on the sums:
//   EPI_LINEAR    C = act(A.op(B) + bias) (+C)
//   EPI_GRU/LSTM  fused forward cell (PyTorch gate order; n = tanh(x_n + r*(W_hn h + b_hn)))
//   EPI_*_BWD     dh_{t-1} = dG_h W_hh + carry, then the previous step's cell backward
// so that one training-size recurrent timestep is ONE launch, forward and backward.
//
// Lane maps of v_mfma_f32_16x16x4_f32: A[i = lane&15][k = lane>>4], B[k = lane>>4][j = lane&15],
// D: col = lane&15, row = 4*(lane>>4) + reg.  Each lane loads 4 consecutive k (one float4)
// of its row; MFMA s of the 4 pairs element s of A's and B's float4, i.e. k = 4*(lane>>4)+s
// on both sides, so the four MFMAs of a K block cover its 16 k values exactly once.
#include "common.hpp"
#include "cell_math.hpp"

namespace cvae {

constexpr int SK_U = 4;    // K blocks in flight per wave

struct SkinnyBatch { SkinnyArgs a[SKINNY_MAX_BATCH]; };

template <int MT, int NG, int EPI, bool FAST>
__global__ __launch_bounds__((MT * NG >= 4) ? 512 : 1024) void skinny_kernel(SkinnyBatch batch, int vecA, int vecB) {
  const SkinnyArgs& p = batch.a[blockIdx.z];
  extern __shared__ __attribute__((aligned(16))) float red[];   // [NW][MT*NG][64 lanes][4]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, NW = blockDim.x >> 6;
  const int n0 = blockIdx.x * 16, m0 = blockIdx.y * (16 * MT);
  const int li = lane & 15, lq = lane >> 4;
  constexpr bool FWD_CELL = (EPI == EPI_GRU || EPI == EPI_LSTM);
  constexpr bool BWD_CELL = (EPI == EPI_GRU_BWD || EPI == EPI_LSTM_BWD);
  constexpr int G = (EPI == EPI_GRU || EPI == EPI_GRU_BWD) ? 3 : 4;

  // ---- epilogue operands of this thread's first output, fetched ahead of the main loop
  const int e_mt = tid >> 8, e_ml = (tid & 255) >> 4, e_j = tid & 15;
  const int e_m = m0 + e_mt * 16 + e_ml, e_n = n0 + e_j;
  const bool e_ok = tid < MT * 256 && e_m < p.M && e_n < p.N;
  float pre_x[4] = {0.f, 0.f, 0.f, 0.f}, pre_b[4] = {0.f, 0.f, 0.f, 0.f}, pre_h = 0.f, pre_c = 0.f;
  float pre_g[4] = {0.f, 0.f, 0.f, 0.f}, pre_c2 = 0.f, pre_d = 0.f, pre_dc = 0.f;
  bool pre_live = true;
  if (e_ok) {
    if constexpr (FWD_CELL) {
      const int H = p.N;
      const float* xp = p.xproj + (long)(e_m % p.xmod) * p.ldx + e_n;
      const float* xa = p.xadd ? p.xadd + (long)e_m * (NG * H) + e_n : nullptr;
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        pre_x[g] = xp[g * H] + (xa ? xa[g * H] : 0.f);
        pre_b[g] = p.b_hh[g * H + e_n];
      }
      pre_h = p.h_prev[(long)e_m * H + e_n];
      if constexpr (EPI == EPI_LSTM) pre_c = p.c_prev[(long)e_m * H + e_n];
      pre_live = (p.lengths == nullptr) || ((int64_t)p.t < p.lengths[e_m]);
    } else if constexpr (BWD_CELL) {
      const int H = p.N;
      const long o = (long)e_m * H + e_n;
      pre_d = p.dh_buf[o] + ((p.has_prev && p.dhs_prev) ? p.dhs_prev[o] : 0.f);
      if (p.has_prev) {
        const float* gp = p.gates_prev + (long)e_m * 4 * H + e_n;
#pragma unroll
        for (int g = 0; g < 4; ++g) pre_g[g] = gp[g * H];
        pre_h = p.hin_prev[o];
        if constexpr (EPI == EPI_LSTM_BWD) { pre_c = p.cin_prev[o]; pre_c2 = p.cout_prev[o]; pre_dc = p.dc_buf[o]; }
        pre_live = (p.lengths == nullptr) || ((int64_t)p.t_prev < p.lengths[e_m]);
      }
    } else {
      pre_b[0] = p.bias ? p.bias[e_n] : 0.f;
      if (p.beta) pre_h = p.C[(long)e_m * p.ldc + e_n];
    }
  }

  f32x4 acc[MT][NG];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int g = 0; g < NG; ++g) acc[i][g] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nkb = (p.K + 15) >> 4;
  const int ncol = n0 + li;
  if constexpr (FAST) {
    // Host-checked: every row / column / k of this block is in range, operands 16-B aligned,
    // transB, nkb % NW == 0.  No guards: SK_U K blocks of loads are in flight before the MFMAs.
    const float* Ap[MT];
    const float* Bp[NG];
#pragma unroll
    for (int i = 0; i < MT; ++i) Ap[i] = p.A + (long)(m0 + i * 16 + li) * p.lda + 4 * lq;
#pragma unroll
    for (int g = 0; g < NG; ++g) Bp[g] = p.B + ((long)ncol + g * p.gstride) * p.ldb + 4 * lq;
    const int iters = nkb / NW;
    int it = 0;
    for (; it + SK_U <= iters; it += SK_U) {
      float4 a[SK_U][MT], b[SK_U][NG];
#pragma unroll
      for (int u = 0; u < SK_U; ++u) {
        const int k = (wave + (it + u) * NW) * 16;
#pragma unroll
        for (int i = 0; i < MT; ++i) a[u][i] = *reinterpret_cast<const float4*>(Ap[i] + k);
#pragma unroll
        for (int g = 0; g < NG; ++g) b[u][g] = *reinterpret_cast<const float4*>(Bp[g] + k);
      }
#pragma unroll
      for (int u = 0; u < SK_U; ++u)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int g = 0; g < NG; ++g) {
            acc[i][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][i].x, b[u][g].x, acc[i][g], 0, 0, 0);
            acc[i][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][i].y, b[u][g].y, acc[i][g], 0, 0, 0);
            acc[i][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][i].z, b[u][g].z, acc[i][g], 0, 0, 0);
            acc[i][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][i].w, b[u][g].w, acc[i][g], 0, 0, 0);
          }
    }
    for (; it < iters; ++it) {
      const int k = (wave + it * NW) * 16;
      float4 a[MT], b[NG];
#pragma unroll
      for (int i = 0; i < MT; ++i) a[i] = *reinterpret_cast<const float4*>(Ap[i] + k);
#pragma unroll
      for (int g = 0; g < NG; ++g) b[g] = *reinterpret_cast<const float4*>(Bp[g] + k);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          acc[i][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].x, b[g].x, acc[i][g], 0, 0, 0);
          acc[i][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].y, b[g].y, acc[i][g], 0, 0, 0);
          acc[i][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].z, b[g].z, acc[i][g], 0, 0, 0);
          acc[i][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].w, b[g].w, acc[i][g], 0, 0, 0);
        }
    }
  } else {
    constexpr int GEN_U = 1;
    for (int kb0 = wave; kb0 < nkb; kb0 += NW * GEN_U) {
      float4 a[GEN_U][MT], b[GEN_U][NG];
  #pragma unroll
      for (int u = 0; u < GEN_U; ++u) {
        const int kb = kb0 + u * NW;
        const int k = kb * 16 + 4 * lq;
  #pragma unroll
        for (int i = 0; i < MT; ++i) {
          const int row = m0 + i * 16 + li;
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
          if (kb < nkb && row < p.M) {
            const float* src = p.A + (long)row * p.lda + k;
            if (vecA && k + 3 < p.K) {
              v = *reinterpret_cast<const float4*>(src);
            } else {
              if (k + 0 < p.K) v.x = src[0];
              if (k + 1 < p.K) v.y = src[1];
              if (k + 2 < p.K) v.z = src[2];
              if (k + 3 < p.K) v.w = src[3];
            }
          }
          a[u][i] = v;
        }
  #pragma unroll
        for (int g = 0; g < NG; ++g) {
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
          if (kb < nkb && ncol < p.N) {
            const long col = ncol + g * p.gstride;
            if (p.transB) {
              const float* src = p.B + col * p.ldb + k;
              if (vecB && k + 3 < p.K) {
                v = *reinterpret_cast<const float4*>(src);
              } else {
                if (k + 0 < p.K) v.x = src[0];
                if (k + 1 < p.K) v.y = src[1];
                if (k + 2 < p.K) v.z = src[2];
                if (k + 3 < p.K) v.w = src[3];
              }
            } else {
              if (k + 0 < p.K) v.x = p.B[(long)(k + 0) * p.ldb + col];
              if (k + 1 < p.K) v.y = p.B[(long)(k + 1) * p.ldb + col];
              if (k + 2 < p.K) v.z = p.B[(long)(k + 2) * p.ldb + col];
              if (k + 3 < p.K) v.w = p.B[(long)(k + 3) * p.ldb + col];
            }
          }
          b[u][g] = v;
        }
      }
  #pragma unroll
      for (int u = 0; u < GEN_U; ++u) {
  #pragma unroll
        for (int i = 0; i < MT; ++i)
  #pragma unroll
          for (int g = 0; g < NG; ++g) {
            acc[i][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][i].x, b[u][g].x, acc[i][g], 0, 0, 0);
            acc[i][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][i].y, b[u][g].y, acc[i][g], 0, 0, 0);
            acc[i][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][i].z, b[u][g].z, acc[i][g], 0, 0, 0);
            acc[i][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][i].w, b[u][g].w, acc[i][g], 0, 0, 0);
          }
      }
    }

  }

#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int g = 0; g < NG; ++g)
      *reinterpret_cast<f32x4*>(red + ((wave * (MT * NG) + i * NG + g) * 64 + lane) * 4) = acc[i][g];
  __syncthreads();

  for (int idx = tid; idx < MT * 256; idx += blockDim.x) {
    const int mt = idx >> 8, rem = idx & 255, ml = rem >> 4, j = rem & 15;
    const int m = m0 + mt * 16 + ml, n = n0 + j;
    if (m >= p.M || n >= p.N) continue;
    const bool first = (idx == tid);       // prefetched operands are valid for the first output only
    const int src = (j + 16 * (ml >> 2)) * 4 + (ml & 3);
    float sum[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      float sacc = 0.f;
      for (int w = 0; w < NW; ++w) sacc += red[(w * (MT * NG) + mt * NG + g) * 256 + src];
      sum[g] = sacc;
    }
    if constexpr (EPI == EPI_LINEAR) {
      float v = sum[0] + (first ? pre_b[0] : (p.bias ? p.bias[n] : 0.f));
      v = apply_act(v, p.act);
      float* c = p.C + (long)m * p.ldc + n;
      if (p.beta) v += first ? pre_h : *c;
      *c = v;
    } else if constexpr (FWD_CELL) {
      const int H = p.N;
      float x[NG], bh[NG], hprev, cprev = 0.f;
      bool live;
      if (first) {
#pragma unroll
        for (int g = 0; g < NG; ++g) { x[g] = pre_x[g]; bh[g] = pre_b[g]; }
        hprev = pre_h; cprev = pre_c; live = pre_live;
      } else {
        const float* xp = p.xproj + (long)(m % p.xmod) * p.ldx + n;
        const float* xa = p.xadd ? p.xadd + (long)m * (NG * H) + n : nullptr;
#pragma unroll
        for (int g = 0; g < NG; ++g) { x[g] = xp[g * H] + (xa ? xa[g * H] : 0.f); bh[g] = p.b_hh[g * H + n]; }
        hprev = p.h_prev[(long)m * H + n];
        if constexpr (EPI == EPI_LSTM) cprev = p.c_prev[(long)m * H + n];
        live = (p.lengths == nullptr) || ((int64_t)p.t < p.lengths[m]);
      }
      if constexpr (EPI == EPI_GRU) {
        const float hn = sum[2] + bh[2];
        const float r = sigmoidf_(x[0] + sum[0] + bh[0]);
        const float z = sigmoidf_(x[1] + sum[1] + bh[1]);
        const float nn = tanhf_(x[2] + r * hn);
        p.h_out[(long)m * H + n] = live ? (1.f - z) * nn + z * hprev : hprev;
        if (p.gates) {
          float* gp = p.gates + (long)m * p.ldg + n;
          gp[0] = r; gp[H] = z; gp[2 * H] = nn; gp[3 * H] = hn;
        }
      } else {
        const float ig = sigmoidf_(x[0] + sum[0] + bh[0]);
        const float fg = sigmoidf_(x[1] + sum[1] + bh[1]);
        const float gg = tanhf_(x[2] + sum[2] + bh[2]);
        const float og = sigmoidf_(x[3] + sum[3] + bh[3]);
        const float cnew = fg * cprev + ig * gg;
        p.h_out[(long)m * H + n] = live ? og * tanhf_(cnew) : hprev;
        p.c_out[(long)m * H + n] = live ? cnew : cprev;
        if (p.gates) {
          float* gp = p.gates + (long)m * p.ldg + n;
          gp[0] = ig; gp[H] = fg; gp[2 * H] = gg; gp[3 * H] = og;
        }
      }
    } else {   // BWD_CELL
      const int H = p.N;
      const long o = (long)m * H + n;
      float dh = sum[0];
      if (!p.has_prev) {
        p.dh_buf[o] = dh + (first ? pre_d : p.dh_buf[o]);
      } else {
        float gv[4], hin, cin = 0.f, cout = 0.f, dc = 0.f;
        bool live;
        if (first) {
          dh += pre_d;
#pragma unroll
          for (int g = 0; g < 4; ++g) gv[g] = pre_g[g];
          hin = pre_h; cin = pre_c; cout = pre_c2; dc = pre_dc; live = pre_live;
        } else {
          dh += p.dh_buf[o] + (p.dhs_prev ? p.dhs_prev[o] : 0.f);
          const float* gp = p.gates_prev + (long)m * 4 * H + n;
#pragma unroll
          for (int g = 0; g < 4; ++g) gv[g] = gp[g * H];
          hin = p.hin_prev[o];
          if constexpr (EPI == EPI_LSTM_BWD) { cin = p.cin_prev[o]; cout = p.cout_prev[o]; dc = p.dc_buf[o]; }
          live = (p.lengths == nullptr) || ((int64_t)p.t_prev < p.lengths[m]);
        }
        // cell_bwd_point reads the activations with stride H: stage them with stride 1
        float act[4] = {gv[0], gv[1], gv[2], gv[3]};
        float ox[4], oh[4];
        constexpr int CELL = (EPI == EPI_GRU_BWD) ? CELL_GRU : CELL_LSTM;
        const float direct = cell_bwd_point<CELL>(live, dh, dc, act, 1, hin, cin, cout, ox, oh);
        float* px = p.dgx_prev + (long)m * G * H + n;
#pragma unroll
        for (int g = 0; g < G; ++g) px[g * H] = ox[g];
        if (p.dgh_prev) {
          float* ph = p.dgh_prev + (long)m * G * H + n;
#pragma unroll
          for (int g = 0; g < G; ++g) ph[g * H] = oh[g];
        }
        p.dh_buf[o] = direct;
        if constexpr (EPI == EPI_LSTM_BWD) p.dc_buf[o] = dc;
      }
    }
  }
}

template <int MT, bool FAST>
static int launch_mt(const SkinnyBatch& a, int epi, int nw, dim3 grid, size_t lds, int vecA, int vecB,
                     hipStream_t s) {
  dim3 block(64 * nw);
  switch (epi) {
    case EPI_LINEAR: hipLaunchKernelGGL((skinny_kernel<MT, 1, EPI_LINEAR, FAST>), grid, block, lds, s, a, vecA, vecB); break;
    case EPI_GRU: hipLaunchKernelGGL((skinny_kernel<MT, 3, EPI_GRU, FAST>), grid, block, lds, s, a, vecA, vecB); break;
    case EPI_LSTM: hipLaunchKernelGGL((skinny_kernel<MT, 4, EPI_LSTM, FAST>), grid, block, lds, s, a, vecA, vecB); break;
    case EPI_GRU_BWD: hipLaunchKernelGGL((skinny_kernel<MT, 1, EPI_GRU_BWD, FAST>), grid, block, lds, s, a, vecA, vecB); break;
    default: hipLaunchKernelGGL((skinny_kernel<MT, 1, EPI_LSTM_BWD, FAST>), grid, block, lds, s, a, vecA, vecB); break;
  }
  CVAE_LAUNCH_CHECK("skinny_kernel");
  return 0;
}

static int check_one(const SkinnyArgs& a, int epi, int ng) {
  CVAE_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0, "skinny: empty problem");
  CVAE_REQUIRE(a.A && a.B, "skinny: null operand");
  CVAE_REQUIRE(a.ngroups == ng, "skinny: ngroups does not match the epilogue");
  if (epi == EPI_LINEAR) {
    CVAE_REQUIRE(a.C && a.ldc >= a.N, "skinny: bad C");
  } else if (epi == EPI_GRU || epi == EPI_LSTM) {
    CVAE_REQUIRE(a.xproj && a.b_hh && a.h_prev && a.h_out && a.xmod > 0, "skinny: cell epilogue operands missing");
    CVAE_REQUIRE(epi != EPI_LSTM || (a.c_prev && a.c_out), "skinny: LSTM needs c_prev / c_out");
    CVAE_REQUIRE(a.gstride == a.N, "skinny: gate stride must equal H");
  } else {
    CVAE_REQUIRE(a.dh_buf && a.transB, "skinny: backward epilogue needs dh_buf and W_hh^T");
    CVAE_REQUIRE(!a.has_prev || (a.gates_prev && a.hin_prev && a.dgx_prev), "skinny: backward epilogue operands missing");
    CVAE_REQUIRE(!a.has_prev || epi != EPI_GRU_BWD || a.dgh_prev, "skinny: GRU backward needs dgh_prev");
    CVAE_REQUIRE(epi != EPI_LSTM_BWD || !a.has_prev || (a.cin_prev && a.cout_prev && a.dc_buf),
                 "skinny: LSTM backward needs cell states and dc_buf");
  }
  return 0;
}

int launch_skinny(const SkinnyArgs& a, int epi, hipStream_t s) { return launch_skinny_multi(&a, 1, epi, s); }

int launch_skinny_multi(const SkinnyArgs* arr, int n, int epi, hipStream_t s) {
  CVAE_REQUIRE(n >= 1 && n <= SKINNY_MAX_BATCH, "skinny: batch of %d problems (max %d)", n, SKINNY_MAX_BATCH);
  CVAE_REQUIRE(epi >= EPI_LINEAR && epi <= EPI_LSTM_BWD, "skinny: bad epilogue");
  const int ng = epi == EPI_GRU ? 3 : (epi == EPI_LSTM ? 4 : 1);
  SkinnyBatch batch = {};
  const SkinnyArgs& a = arr[0];
  int vecA = 1, vecB = a.transB;
  for (int i = 0; i < n; ++i) {
    CVAE_TRY(check_one(arr[i], epi, ng));
    CVAE_REQUIRE(arr[i].M == a.M && arr[i].N == a.N && arr[i].K == a.K && arr[i].transB == a.transB &&
                 arr[i].lda == a.lda && arr[i].ldb == a.ldb, "skinny: batched problems differ in shape");
    vecA = vecA && aligned16(arr[i].A) && (arr[i].lda % 4 == 0);
    vecB = vecB && aligned16(arr[i].B) && (arr[i].ldb % 4 == 0);
    batch.a[i] = arr[i];
  }
  // rows per block: as few as possible while the grid is small (the step is bounded by the MFMA rate of
  // the CUs it occupies), more rows per block (fewer re-reads of op(B)) once the grid fills the chip
  int mt = a.M <= 16 ? 1 : (a.M <= 32 ? 2 : 4);
  while (mt > 1 && (long)n * cdiv(a.N, 16) * cdiv(a.M, 16 * mt) < 256) mt >>= 1;
  const int nkb = (a.K + 15) / 16;
  int nw = 64 / (mt * ng);            // keep the reduction buffer <= 64 KiB
  const int maxw = (mt * ng >= 4) ? 8 : 16;     // matches the kernel's __launch_bounds__
  if (nw > maxw) nw = maxw;
  if (nw > nkb) nw = nkb;
  if (nw < 1) nw = 1;
  dim3 grid(cdiv(a.N, 16), cdiv(a.M, 16 * mt), n);
  ProfScope prof(PK_SKINNY, s, 2.0 * a.M * a.N * a.K * ng * n);
  // guard-free kernel when the block tiling covers the problem exactly
  bool fast = a.transB && vecA && vecB && (a.M % (16 * mt) == 0) && (a.N % 16 == 0) && (a.K % 16 == 0);
  if (fast) {
    while (nw > 1 && nkb % nw != 0) --nw;          // every wave gets the same number of K blocks
    fast = (nkb % nw == 0);
  }
  const size_t lds_f = (size_t)nw * mt * ng * 256 * sizeof(float);
  if (fast) {
    if (mt == 1) return launch_mt<1, true>(batch, epi, nw, grid, lds_f, vecA, vecB, s);
    if (mt == 2) return launch_mt<2, true>(batch, epi, nw, grid, lds_f, vecA, vecB, s);
    return launch_mt<4, true>(batch, epi, nw, grid, lds_f, vecA, vecB, s);
  }
  if (mt == 1) return launch_mt<1, false>(batch, epi, nw, grid, lds_f, vecA, vecB, s);
  if (mt == 2) return launch_mt<2, false>(batch, epi, nw, grid, lds_f, vecA, vecB, s);
  return launch_mt<4, false>(batch, epi, nw, grid, lds_f, vecA, vecB, s);
}

}  // namespace cvae
