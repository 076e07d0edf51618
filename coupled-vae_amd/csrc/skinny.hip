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
#include <stdlib.h>
#include "cell_math.hpp"

namespace cvae {

constexpr int SK_U = 4;    // K blocks in flight per wave

struct SkinnyBatch { SkinnyArgs a[SKINNY_MAX_BATCH]; };

struct SkPre { float x[4], b[4], g[4], h, c, c2, d, dc; bool live; };

// Epilogue operands of this thread's first output, fetched ahead of the main loop.
template <int MT, int NG, int EPI>
__device__ __forceinline__ void sk_prefetch(const SkinnyArgs& p, int tid, int m0, int n0, SkPre& q) {
  constexpr bool FWD_CELL = (EPI == EPI_GRU || EPI == EPI_LSTM);
  constexpr bool BWD_CELL = (EPI == EPI_GRU_BWD || EPI == EPI_LSTM_BWD);
  const int e_mt = tid >> 8, e_ml = (tid & 255) >> 4, e_j = tid & 15;
  const int e_m = m0 + e_mt * 16 + e_ml, e_n = n0 + e_j;
  const bool e_ok = tid < MT * 256 && e_m < p.M && e_n < p.N;
#pragma unroll
  for (int g = 0; g < 4; ++g) { q.x[g] = 0.f; q.b[g] = 0.f; q.g[g] = 0.f; }
  q.h = q.c = q.c2 = q.d = q.dc = 0.f;
  q.live = true;
  if (e_ok) {
    if constexpr (FWD_CELL) {
      const int H = p.N;
      const float* xp = p.xproj + (long)(e_m % p.xmod) * p.ldx + e_n;
      const float* xa = p.xadd ? p.xadd + (long)e_m * (NG * H) + e_n : nullptr;
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        q.x[g] = xp[g * H] + (xa ? xa[g * H] : 0.f);
        q.b[g] = p.b_hh[g * H + e_n];
      }
      q.h = p.h_prev[(long)e_m * H + e_n];
      if constexpr (EPI == EPI_LSTM) q.c = p.c_prev[(long)e_m * H + e_n];
      q.live = (p.lengths == nullptr) || ((int64_t)p.t < p.lengths[e_m]);
    } else if constexpr (BWD_CELL) {
      const int H = p.N;
      const long o = (long)e_m * H + e_n;
      q.d = p.dh_buf[o] + ((p.has_prev && p.dhs_prev) ? p.dhs_prev[o] : 0.f);
      if (p.has_prev) {
        const float* gp = p.gates_prev + (long)e_m * 4 * H + e_n;
#pragma unroll
        for (int g = 0; g < 4; ++g) q.g[g] = gp[g * H];
        q.h = p.hin_prev[o];
        if constexpr (EPI == EPI_LSTM_BWD) { q.c = p.cin_prev[o]; q.c2 = p.cout_prev[o]; q.dc = p.dc_buf[o]; }
        q.live = (p.lengths == nullptr) || ((int64_t)p.t_prev < p.lengths[e_m]);
      }
    } else {
      q.b[0] = p.bias ? p.bias[e_n] : 0.f;
      if (p.beta) q.h = p.C[(long)e_m * p.ldc + e_n];
    }
  }

}

// Sums the NW waves' partial accumulators (red: [NW][MT*NG][64 lanes][4]) and runs the epilogue on the sums.
template <int MT, int NG, int EPI>
__device__ __forceinline__ void sk_epilogue(const SkinnyArgs& p, const float* red, int NW, int tid, int nthreads,
                                            int m0, int n0, const SkPre& q) {
  constexpr bool FWD_CELL = (EPI == EPI_GRU || EPI == EPI_LSTM);
  constexpr int G = (EPI == EPI_GRU || EPI == EPI_GRU_BWD) ? 3 : 4;
  for (int idx = tid; idx < MT * 256; idx += nthreads) {
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
      float v = sum[0] + (first ? q.b[0] : (p.bias ? p.bias[n] : 0.f));
      v = apply_act(v, p.act);
      float* c = p.C + (long)m * p.ldc + n;
      if (p.beta) v += first ? q.h : *c;
      *c = v;
    } else if constexpr (FWD_CELL) {
      const int H = p.N;
      float x[NG], bh[NG], hprev, cprev = 0.f;
      bool live;
      if (first) {
#pragma unroll
        for (int g = 0; g < NG; ++g) { x[g] = q.x[g]; bh[g] = q.b[g]; }
        hprev = q.h; cprev = q.c; live = q.live;
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
        p.dh_buf[o] = dh + (first ? q.d : p.dh_buf[o]);
      } else {
        float gv[4], hin, cin = 0.f, cout = 0.f, dc = 0.f;
        bool live;
        if (first) {
          dh += q.d;
#pragma unroll
          for (int g = 0; g < 4; ++g) gv[g] = q.g[g];
          hin = q.h; cin = q.c; cout = q.c2; dc = q.dc; live = q.live;
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

template <int MT, int NG, int EPI, bool FAST>
__global__ __launch_bounds__((MT * NG >= 4) ? 512 : 1024) void skinny_kernel(SkinnyBatch batch, int vecA, int vecB) {
  const SkinnyArgs& p = batch.a[blockIdx.z];
  extern __shared__ __attribute__((aligned(16))) float red[];   // [NW][MT*NG][64 lanes][4]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, NW = blockDim.x >> 6;
  const int n0 = blockIdx.x * 16, m0 = blockIdx.y * (16 * MT);
  const int li = lane & 15, lq = lane >> 4;

  SkPre q;
  sk_prefetch<MT, NG, EPI>(p, tid, m0, n0, q);

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

  sk_epilogue<MT, NG, EPI>(p, red, NW, tid, blockDim.x, m0, n0, q);
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


// ------------------------------------------------------------------ weight-streaming split-bf16 variant
// (common.hpp: launch_skinny_stream).  The fp32-input kernel above is bound by its own MFMAs once the weights
// are large (64 rows x 2048 units x 4 gates x K = 2048: 27 us of v_mfma_f32_16x16x4_f32 per CU) and reads them in
// 64-byte pieces; here the weights arrive as 1 KB wave-instructions from an image laid out in fragment order,
// the activations are split into bf16 hi / lo in registers, and three v_mfma_f32_16x16x32_bf16 per fragment pair
// (hi.hi, hi.lo, lo.hi) accumulate in fp32: the step is bound by streaming the weights (Infinity Cache / HBM).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
constexpr int ST_NW = 8;          // waves per block: K blocks (32 wide) w, w+8, ... of the block's K range

struct SkImgs { const void* p[SKINNY_MAX_BATCH]; };

// image slot (16 bytes) of (column block bx, K block kb, group g, half, lane): ((bx*nkb + kb)*NG + g)*128 + half*64 + lane;
// lane (j = lane & 15, q = lane >> 4) holds k = 32*kb + 8*q .. +7 of column bx*cb + g*gs + j
__global__ __launch_bounds__(256) void stream_presplit_kernel(const float* __restrict__ W, long col_stride, long k_stride,
                                                              int nkb, int ng, int cb, long gs, long total,
                                                              bf16x8* __restrict__ img) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;        // (bx, kb, g, lane)
  if (idx >= total) return;
  const int lane = (int)(idx & 63);
  const long t = idx >> 6;
  const int g = (int)(t % ng);
  const long t2 = t / ng;
  const int kb = (int)(t2 % nkb);
  const long bx = t2 / nkb;
  const long col = bx * cb + g * gs + (lane & 15);
  const long k = (long)kb * 32 + 8 * (lane >> 4);
  const float* src = W + col * col_stride + k * k_stride;
  f32x8 v;
  if (k_stride == 1) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(src), b = *reinterpret_cast<const f32x4*>(src + 4);
    v = f32x8{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = src[e * k_stride];
  }
  const bf16x8 hi = __builtin_convertvector(v, bf16x8);
  const bf16x8 lo = __builtin_convertvector(v - __builtin_convertvector(hi, f32x8), bf16x8);
  bf16x8* dst = img + t * 128 + lane;
  dst[0] = hi;
  dst[64] = lo;
}

template <int MT, int NG, int EPI>
__global__ __launch_bounds__(64 * ST_NW) void skinny_stream_kernel(SkinnyBatch batch, SkImgs imgs, int nkb, int kb_split,
                                                                  float* __restrict__ parts) {
  const SkinnyArgs& p = batch.a[blockIdx.z];
  extern __shared__ __attribute__((aligned(16))) float red[];   // [ST_NW][MT*NG][64 lanes][4]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lq = lane >> 4;
  constexpr bool CELL_EPI = (EPI != EPI_LINEAR);
  const int n0 = blockIdx.x * (CELL_EPI ? 16 : 16 * NG);

  SkPre q;
  if constexpr (CELL_EPI) sk_prefetch<MT, NG, EPI>(p, tid, 0, n0, q);

  f32x4 acc[MT][NG];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int g = 0; g < NG; ++g) acc[i][g] = f32x4{0.f, 0.f, 0.f, 0.f};

  // Row tiles that hold no live row contribute nothing the epilogue uses (forward: a dead row only carries its
  // state; partial-product form: a dead row's A is exactly zero and so is its output, written below): neither
  // loaded nor multiplied.  bound = 1 + the highest row with t < lengths[row] (rows sorted by length: the live count).
  int mt_live = MT;
  if (p.lengths) {
    const bool lv = lane < p.M && (int64_t)p.t < p.lengths[lane];
    const unsigned long long mask = __ballot(lv);
    const int bound = mask ? 64 - __clzll(mask) : 0;
    mt_live = (bound + 15) >> 4;
  }
  const int kb0 = blockIdx.y * kb_split;
  const int kb1 = (kb0 + kb_split < nkb && mt_live > 0) ? kb0 + kb_split : (mt_live > 0 ? nkb : kb0);
  const bf16x8* bimg = reinterpret_cast<const bf16x8*>(imgs.p[blockIdx.z]) + (long)blockIdx.x * nkb * (NG * 128) + lane;
  const float* Ap[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    int row = i * 16 + li;
    row = row < p.M ? row : p.M - 1;                 // rows past M: a valid address, result discarded
    Ap[i] = p.A + (long)row * p.lda + 8 * lq;
  }

  f32x4 ar[2][MT][2];
  bf16x8 bh[2][NG], bl[2][NG];
#define ST_LOAD(set, kb)                                                                   \
  do {                                                                                     \
    const bf16x8* bsrc = bimg + (long)(kb) * (NG * 128);                                   \
    _Pragma("unroll") for (int g = 0; g < NG; ++g) {                                       \
      bh[set][g] = bsrc[g * 128];                                                          \
      bl[set][g] = bsrc[g * 128 + 64];                                                     \
    }                                                                                      \
    _Pragma("unroll") for (int i = 0; i < MT; ++i) {                                       \
      if (i < mt_live) {                                                                   \
        ar[set][i][0] = *reinterpret_cast<const f32x4*>(Ap[i] + (long)(kb) * 32);          \
        ar[set][i][1] = *reinterpret_cast<const f32x4*>(Ap[i] + (long)(kb) * 32 + 4);      \
      }                                                                                    \
    }                                                                                      \
  } while (0)

  int kb = kb0 + wave;
  if (kb < kb1) ST_LOAD(0, kb);
  for (; kb < kb1; kb += 2 * ST_NW) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int kc = kb + ST_NW * u;
      if (kc < kb1) {
        if (kc + ST_NW < kb1) ST_LOAD(u ^ 1, kc + ST_NW);
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          if (i < mt_live) {                                 // wave-uniform
            const f32x4 a0 = ar[u][i][0], a1 = ar[u][i][1];
            const f32x8 v = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
            const bf16x8 ah = __builtin_convertvector(v, bf16x8);
            const bf16x8 al = __builtin_convertvector(v - __builtin_convertvector(ah, f32x8), bf16x8);
#pragma unroll
            for (int g = 0; g < NG; ++g) acc[i][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[u][g], acc[i][g], 0, 0, 0);
#pragma unroll
            for (int g = 0; g < NG; ++g) acc[i][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[u][g], acc[i][g], 0, 0, 0);
#pragma unroll
            for (int g = 0; g < NG; ++g) acc[i][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[u][g], acc[i][g], 0, 0, 0);
          }
        }
      }
    }
  }
#undef ST_LOAD

#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int g = 0; g < NG; ++g)
      *reinterpret_cast<f32x4*>(red + ((wave * (MT * NG) + i * NG + g) * 64 + lane) * 4) = acc[i][g];
  __syncthreads();

  if constexpr (CELL_EPI) {
    sk_epilogue<MT, NG, EPI>(p, red, ST_NW, tid, 64 * ST_NW, 0, n0, q);
  } else {
    float* out = parts + ((long)blockIdx.z * gridDim.y + blockIdx.y) * p.M * p.N;
    for (int idx = tid; idx < MT * 256; idx += 64 * ST_NW) {
      const int mt = idx >> 8, rem = idx & 255, ml = rem >> 4, j = rem & 15;
      const int m = mt * 16 + ml;
      if (m >= p.M) continue;
      const int src = (j + 16 * (ml >> 2)) * 4 + (ml & 3);
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        float sacc = 0.f;
#pragma unroll
        for (int w = 0; w < ST_NW; ++w) sacc += red[(w * (MT * NG) + mt * NG + g) * 256 + src];
        out[(long)m * p.N + n0 + 16 * g + j] = sacc;
      }
    }
  }
}

static long g_stream_min_bytes = -1;
long stream_rnn_min_weight_bytes(long set_to) {
  if (g_stream_min_bytes < 0) {
    const char* e = getenv("CVAE_STREAM_RNN_MIN_MB");       // 0: every supported shape; default 8 MB of W_hh
    g_stream_min_bytes = (e ? atol(e) : 8) << 20;
  }
  const long prev = g_stream_min_bytes;
  if (set_to >= 0) g_stream_min_bytes = set_to;
  return prev;
}

bool stream_rnn_supported(int R, int H, int G) {
  if (!split_bf16_enabled() || R < 1 || R > 64 || H % 64 != 0) return false;
  return (long)G * H * H * 4 >= stream_rnn_min_weight_bytes(-1);
}

size_t stream_image_bytes(int ncols, int K) { return (size_t)ncols * K * 4; }

int launch_stream_presplit(const float* W, long col_stride, long k_stride, int ncols, int K, int ng, int cb, long gs,
                           void* img, hipStream_t s) {
  CVAE_REQUIRE(W && img && aligned16(img) && K % 32 == 0 && ncols % (16 * ng) == 0 && ng >= 1 && ng <= 4,
               "stream_presplit: bad argument (ncols=%d K=%d ng=%d)", ncols, K, ng);
  CVAE_REQUIRE(k_stride != 1 || (aligned16(W) && col_stride % 4 == 0), "stream_presplit: unaligned weights");
  const long total = (long)(ncols / 16) * (K / 32) * 64;    // (bx, g) pairs = ncols / 16
  hipLaunchKernelGGL(stream_presplit_kernel, dim3(cdiv(total, 256)), dim3(256), 0, s, W, col_stride, k_stride, K / 32, ng,
                     cb, gs, total, reinterpret_cast<bf16x8*>(img));
  CVAE_LAUNCH_CHECK("stream_presplit_kernel");
  return 0;
}

// K ranges per problem of the partial-product form: enough blocks to cover the chip, every range a whole number
// of 8-K-block rounds where possible
int stream_linear_ksplit(int n, int N, int K) {
  const int col_blocks = n * (N / 64), nkb = K / 32;
  int ks = 1;
  while (col_blocks * ks < 224 && nkb / (ks * 2) >= ST_NW && ks < 16) ks *= 2;
  return ks;
}

template <int MT, int NG, int EPI>
static int launch_stream_t(const SkinnyBatch& b, const SkImgs& im, int nkb, int kb_split, float* parts, dim3 grid,
                           hipStream_t s) {
  const size_t lds = (size_t)ST_NW * MT * NG * 256 * sizeof(float);
  static bool attr_done = false;
  if (lds > 64 * 1024 && !attr_done) {
    CVAE_TRY(check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(&skinny_stream_kernel<MT, NG, EPI>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds),
                       "hipFuncSetAttribute(skinny_stream_kernel)"));
    attr_done = true;
  }
  hipLaunchKernelGGL((skinny_stream_kernel<MT, NG, EPI>), grid, dim3(64 * ST_NW), lds, s, b, im, nkb, kb_split, parts);
  CVAE_LAUNCH_CHECK("skinny_stream_kernel");
  return 0;
}

template <int MT>
static int launch_stream_mt(const SkinnyBatch& b, const SkImgs& im, int epi, int nkb, int kb_split, float* parts, dim3 grid,
                            hipStream_t s) {
  switch (epi) {
    case EPI_GRU: return launch_stream_t<MT, 3, EPI_GRU>(b, im, nkb, kb_split, parts, grid, s);
    case EPI_LSTM: return launch_stream_t<MT, 4, EPI_LSTM>(b, im, nkb, kb_split, parts, grid, s);
    default: return launch_stream_t<MT, 4, EPI_LINEAR>(b, im, nkb, kb_split, parts, grid, s);
  }
}

int launch_skinny_stream(const SkinnyArgs* arr, void* const* imgs, int n, int epi, int ksplit, float* parts,
                         hipStream_t s) {
  CVAE_REQUIRE(n >= 1 && n <= SKINNY_MAX_BATCH && imgs, "skinny_stream: batch of %d problems (max %d)", n, SKINNY_MAX_BATCH);
  CVAE_REQUIRE(epi == EPI_LINEAR || epi == EPI_GRU || epi == EPI_LSTM, "skinny_stream: bad epilogue %d", epi);
  const SkinnyArgs& a = arr[0];
  SkinnyBatch batch = {};
  SkImgs im = {};
  for (int i = 0; i < n; ++i) {
    const SkinnyArgs& x = arr[i];
    CVAE_REQUIRE(x.M == a.M && x.N == a.N && x.K == a.K && x.lda == a.lda, "skinny_stream: batched problems differ in shape");
    CVAE_REQUIRE(x.M >= 1 && x.M <= 64 && x.K % 32 == 0 && x.K > 0, "skinny_stream: M=%d K=%d unsupported", x.M, x.K);
    CVAE_REQUIRE(x.A && imgs[i] && aligned16(x.A) && aligned16(imgs[i]) && x.lda % 4 == 0, "skinny_stream: unaligned operand");
    if (epi == EPI_LINEAR) {
      CVAE_REQUIRE(x.N % 64 == 0 && parts && ksplit >= 1, "skinny_stream: partial form needs N %% 64 == 0 and a parts buffer");
    } else {
      CVAE_TRY(check_one(x, epi, epi == EPI_GRU ? 3 : 4));
      CVAE_REQUIRE(x.N % 16 == 0 && x.N == x.K, "skinny_stream: cell epilogue needs N == K == H, H %% 16 == 0");
    }
    batch.a[i] = x;
    im.p[i] = imgs[i];
  }
  const int nkb = a.K / 32;
  const int ks = epi == EPI_LINEAR ? ksplit : 1;
  const int kb_split = cdiv(nkb, ks);
  dim3 grid(epi == EPI_LINEAR ? a.N / 64 : a.N / 16, ks, n);
  const int ng = epi == EPI_GRU ? 3 : 4;
  ProfScope prof(PK_SKINNY, s, 2.0 * a.M * a.N * a.K * (epi == EPI_LINEAR ? 1 : ng) * n);
  const int mt = cdiv(a.M, 16);
  if (mt == 1) return launch_stream_mt<1>(batch, im, epi, nkb, kb_split, parts, grid, s);
  if (mt == 2) return launch_stream_mt<2>(batch, im, epi, nkb, kb_split, parts, grid, s);
  if (mt == 3) return launch_stream_mt<3>(batch, im, epi, nkb, kb_split, parts, grid, s);
  return launch_stream_mt<4>(batch, im, epi, nkb, kb_split, parts, grid, s);
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
