// Few-row (M <= 64 per block) fp32 products on v_mfma_f32_16x16x4_f32.
//
// One block owns 16 output columns (times `ngroups` gate groups) and 16*MT rows.  Its NW
// waves split the K dimension: wave w takes the 16-deep K blocks w, w+NW, ...  Operands go
// straight from L2 to registers (they are read once per block; an LDS round trip would be
// pure overhead, cdna_hip_programming.md "GEMV / M <= 16" row), partial accumulators are
// summed through LDS and the epilogue runs on the sums:
//   EPI_LINEAR  C = act(A.op(B) + bias) (+C)
//   EPI_GRU     fused GRU cell   (PyTorch gate order r,z,n; n = tanh(x_n + r*(W_hn h + b_hn)))
//   EPI_LSTM    fused LSTM cell  (i,f,g,o)
// so that one training-size recurrent timestep is ONE launch.
//
// Lane maps of v_mfma_f32_16x16x4_f32: A[i = lane&15][k = lane>>4], B[k = lane>>4][j = lane&15],
// D: col = lane&15, row = 4*(lane>>4) + reg.  Each lane loads 4 consecutive k (one float4)
// of its row; MFMA s of the 4 pairs element s of A's and B's float4, i.e. k = 4*(lane>>4)+s
// on both sides, so the four MFMAs of a K block cover its 16 k values exactly once.
#include "common.hpp"

namespace cvae {

template <int MT, int NG, int EPI>
__global__ __launch_bounds__(512) void skinny_kernel(SkinnyArgs p, int vecA, int vecB) {
  extern __shared__ __attribute__((aligned(16))) float red[];   // [NW][MT*NG][64 lanes][4]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, NW = blockDim.x >> 6;
  const int n0 = blockIdx.x * 16, m0 = blockIdx.y * (16 * MT);
  const int li = lane & 15, lq = lane >> 4;

  f32x4 acc[MT][NG];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int g = 0; g < NG; ++g) acc[i][g] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nkb = (p.K + 15) >> 4;
  const int ncol = n0 + li;
  for (int kb = wave; kb < nkb; kb += NW) {
    const int k = kb * 16 + 4 * lq;
    float a[MT][4], b[NG][4];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int row = m0 + i * 16 + li;
      a[i][0] = a[i][1] = a[i][2] = a[i][3] = 0.f;
      if (row < p.M) {
        const float* src = p.A + (long)row * p.lda + k;
        if (vecA && k + 3 < p.K) {
          const float4 v = *reinterpret_cast<const float4*>(src);
          a[i][0] = v.x; a[i][1] = v.y; a[i][2] = v.z; a[i][3] = v.w;
        } else {
#pragma unroll
          for (int s = 0; s < 4; ++s) if (k + s < p.K) a[i][s] = src[s];
        }
      }
    }
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      b[g][0] = b[g][1] = b[g][2] = b[g][3] = 0.f;
      if (ncol < p.N) {
        const long col = ncol + g * p.gstride;
        if (p.transB) {
          const float* src = p.B + col * p.ldb + k;
          if (vecB && k + 3 < p.K) {
            const float4 v = *reinterpret_cast<const float4*>(src);
            b[g][0] = v.x; b[g][1] = v.y; b[g][2] = v.z; b[g][3] = v.w;
          } else {
#pragma unroll
            for (int s = 0; s < 4; ++s) if (k + s < p.K) b[g][s] = src[s];
          }
        } else {
#pragma unroll
          for (int s = 0; s < 4; ++s) if (k + s < p.K) b[g][s] = p.B[(long)(k + s) * p.ldb + col];
        }
      }
    }
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int g = 0; g < NG; ++g)
          acc[i][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][s], b[g][s], acc[i][g], 0, 0, 0);
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
    const int src = (j + 16 * (ml >> 2)) * 4 + (ml & 3);
    float sum[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      float sacc = 0.f;
      for (int w = 0; w < NW; ++w) sacc += red[(w * (MT * NG) + mt * NG + g) * 256 + src];
      sum[g] = sacc;
    }
    if constexpr (EPI == EPI_LINEAR) {
      float v = sum[0] + (p.bias ? p.bias[n] : 0.f);
      v = apply_act(v, p.act);
      float* c = p.C + (long)m * p.ldc + n;
      if (p.beta) v += *c;
      *c = v;
    } else {
      const int H = p.N;
      const bool live = (p.lengths == nullptr) || ((int64_t)p.t < p.lengths[m]);
      const float* xp = p.xproj + (long)(m % p.xmod) * p.ldx + n;
      const float* xa = p.xadd ? p.xadd + (long)m * (NG * H) + n : nullptr;
      float x[NG];
#pragma unroll
      for (int g = 0; g < NG; ++g) x[g] = xp[g * H] + (xa ? xa[g * H] : 0.f);
      const float hprev = p.h_prev[(long)m * H + n];
      if constexpr (EPI == EPI_GRU) {
        const float hn = sum[2] + p.b_hh[2 * H + n];
        const float r = sigmoidf_(x[0] + sum[0] + p.b_hh[n]);
        const float z = sigmoidf_(x[1] + sum[1] + p.b_hh[H + n]);
        const float nn = tanhf_(x[2] + r * hn);
        const float hnew = live ? (1.f - z) * nn + z * hprev : hprev;
        p.h_out[(long)m * H + n] = hnew;
        if (p.gates) {
          float* gp = p.gates + (long)m * p.ldg + n;
          gp[0] = r; gp[H] = z; gp[2 * H] = nn; gp[3 * H] = hn;
        }
      } else {
        const float ig = sigmoidf_(x[0] + sum[0] + p.b_hh[n]);
        const float fg = sigmoidf_(x[1] + sum[1] + p.b_hh[H + n]);
        const float gg = tanhf_(x[2] + sum[2] + p.b_hh[2 * H + n]);
        const float og = sigmoidf_(x[3] + sum[3] + p.b_hh[3 * H + n]);
        const float cprev = p.c_prev[(long)m * H + n];
        const float cnew = fg * cprev + ig * gg;
        const float hnew = og * tanhf_(cnew);
        p.h_out[(long)m * H + n] = live ? hnew : hprev;
        p.c_out[(long)m * H + n] = live ? cnew : cprev;
        if (p.gates) {
          float* gp = p.gates + (long)m * p.ldg + n;
          gp[0] = ig; gp[H] = fg; gp[2 * H] = gg; gp[3 * H] = og;
        }
      }
    }
  }
}

template <int MT>
static int launch_mt(const SkinnyArgs& a, int epi, int nw, dim3 grid, size_t lds, int vecA, int vecB,
                     hipStream_t s) {
  dim3 block(64 * nw);
  if (epi == EPI_LINEAR) hipLaunchKernelGGL((skinny_kernel<MT, 1, EPI_LINEAR>), grid, block, lds, s, a, vecA, vecB);
  else if (epi == EPI_GRU) hipLaunchKernelGGL((skinny_kernel<MT, 3, EPI_GRU>), grid, block, lds, s, a, vecA, vecB);
  else hipLaunchKernelGGL((skinny_kernel<MT, 4, EPI_LSTM>), grid, block, lds, s, a, vecA, vecB);
  CVAE_LAUNCH_CHECK("skinny_kernel");
  return 0;
}

int launch_skinny(const SkinnyArgs& a, int epi, hipStream_t s) {
  CVAE_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0, "skinny: empty problem");
  CVAE_REQUIRE(a.A && a.B, "skinny: null operand");
  const int ng = epi == EPI_LINEAR ? 1 : (epi == EPI_GRU ? 3 : 4);
  CVAE_REQUIRE(a.ngroups == ng, "skinny: ngroups does not match the epilogue");
  if (epi == EPI_LINEAR) {
    CVAE_REQUIRE(a.C && a.ldc >= a.N, "skinny: bad C");
  } else {
    CVAE_REQUIRE(a.xproj && a.b_hh && a.h_prev && a.h_out && a.xmod > 0, "skinny: cell epilogue operands missing");
    CVAE_REQUIRE(epi != EPI_LSTM || (a.c_prev && a.c_out), "skinny: LSTM needs c_prev / c_out");
    CVAE_REQUIRE(a.gstride == a.N, "skinny: gate stride must equal H");
  }
  const int mt = a.M <= 16 ? 1 : (a.M <= 32 ? 2 : 4);
  const int nkb = (a.K + 15) / 16;
  int nw = 64 / (mt * ng);            // keep the reduction buffer <= 64 KiB
  if (nw > 8) nw = 8;
  if (nw > nkb) nw = nkb;
  if (nw < 1) nw = 1;
  const size_t lds = (size_t)nw * mt * ng * 256 * sizeof(float);
  dim3 grid(cdiv(a.N, 16), cdiv(a.M, 16 * mt));
  ProfScope prof(PK_SKINNY, s, 2.0 * a.M * a.N * a.K * ng);
  const int vecA = aligned16(a.A) && (a.lda % 4 == 0);
  const int vecB = a.transB && aligned16(a.B) && (a.ldb % 4 == 0);
  if (mt == 1) return launch_mt<1>(a, epi, nw, grid, lds, vecA, vecB, s);
  if (mt == 2) return launch_mt<2>(a, epi, nw, grid, lds, vecA, vecB, s);
  return launch_mt<4>(a, epi, nw, grid, lds, vecA, vecB, s);
}

}  // namespace cvae
