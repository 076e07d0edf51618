// Hidden->vocab projection fused with softmax cross-entropy
// (reference: ToVocab + SeqLoss, CV/Modules/vae_couple.py:119, CV/Modules/Losses/SeqLoss.py:36-44).
//
// Forward never materialises the (N,V) logits: each block owns 128 rows and a range of
// vocab tiles; per tile it forms the 128x128 logits block with v_mfma_f32_32x32x2_f32 and
// folds it into per-lane running (max, sum-exp) pairs (online softmax), picking up the target
// logit on the way.  Partial pairs go to a small workspace and a second kernel combines them:
//     loss[n] = (target != pad) * (lse[n] - logit[n, target[n]]).
// Backward (launch_projce_backward) recomputes the logits in row chunks that stay in the
// Infinity Cache: chunk GEMM with the softmax-gradient epilogue, then dfinal = P W and
// dW += P^T final as MFMA GEMMs; the full (N,V) matrix is never formed.
#include "common.hpp"
#include "tile.hpp"

namespace cvae {

constexpr int CE_BM = 128, CE_BN = 128;

struct CeFwdArgs {
  const float* X; long ldx;      // [N,E]
  const float* W; long ldw;      // [V,E]
  const float* bias;             // [V]
  const int64_t* tgt;            // [N]
  int N, E, V;
  int tiles_per_split;           // vocab tiles handled by one blockIdx.x
  float* pm; float* ps;          // [2*nsplit, N] partial max / sum-exp
  float* tgt_logit;              // [N]
  int vecX, vecW;
};

__device__ __forceinline__ void lse_merge(float& m, float& s, float m2, float s2) {
  const float M = fmaxf(m, m2);
  const float a = (m == M) ? 1.f : expf(m - M);     // (m == M) also covers -inf == -inf
  const float b = (m2 == M) ? 1.f : expf(m2 - M);
  s = s * a + s2 * b;
  m = M;
}

__global__ __launch_bounds__(256) void ce_fwd_kernel(CeFwdArgs p) {
  using LA = TileLoader<CE_BM, true>;
  using LB = TileLoader<CE_BN, true>;
  __shared__ __attribute__((aligned(16))) float As[BK * LA::LD];
  __shared__ __attribute__((aligned(16))) float Bs[BK * LB::LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int m0 = blockIdx.y * CE_BM;
  const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * 64;
  const int ntiles = (p.V + CE_BN - 1) / CE_BN;
  const int tile_beg = blockIdx.x * p.tiles_per_split;
  const int tile_end = min(ntiles, tile_beg + p.tiles_per_split);
  const int nk = (p.E + BK - 1) / BK;

  float rm[2][16], rs[2][16];
  int64_t tg[2][16];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      rm[i][r] = -INFINITY; rs[i][r] = 0.f;
      const int m = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
      tg[i][r] = m < p.N ? p.tgt[m] : -1;
    }

  LA la; LB lb;
  for (int tile = tile_beg; tile < tile_end; ++tile) {
    const int n0 = tile * CE_BN;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    la.load(p.X, p.ldx, m0, p.N, 0, p.E, p.vecX, tid);
    lb.load(p.W, p.ldw, n0, p.V, 0, p.E, p.vecW, tid);
    __syncthreads();                 // previous tile's MFMA reads are done
    la.store(As, tid);
    lb.store(Bs, tid);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
      if (kt + 1 < nk) {
        la.load(p.X, p.ldx, m0, p.N, (kt + 1) * BK, p.E, p.vecX, tid);
        lb.load(p.W, p.ldw, n0, p.V, (kt + 1) * BK, p.E, p.vecW, tid);
      }
#pragma unroll
      for (int kk = 0; kk < BK; kk += 2) {
        float a[2], b[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) a[i] = As[(kk + lhi) * LA::LD + wm0 + i * 32 + l31];
#pragma unroll
        for (int j = 0; j < 2; ++j) b[j] = Bs[(kk + lhi) * LB::LD + wn0 + j * 32 + l31];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
      }
      if (kt + 1 < nk) {
        __syncthreads();
        la.store(As, tid);
        lb.store(Bs, tid);
        __syncthreads();
      }
    }
    // fold this 128x128 logits block into the per-lane running pairs
    const int nA = n0 + wn0 + l31, nB = nA + 32;
    const bool vA = nA < p.V, vB = nB < p.V;
    const float bA = vA ? p.bias[nA] : 0.f, bB = vB ? p.bias[nB] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float xA = acc[i][0][r] + bA, xB = acc[i][1][r] + bB;
        if (vA && tg[i][r] == (int64_t)nA) p.tgt_logit[m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi] = xA;
        if (vB && tg[i][r] == (int64_t)nB) p.tgt_logit[m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi] = xB;
        float mx = rm[i][r];
        if (vA) mx = fmaxf(mx, xA);
        if (vB) mx = fmaxf(mx, xB);
        if (mx > -INFINITY) {
          float sacc = (rm[i][r] == mx) ? rs[i][r] : rs[i][r] * expf(rm[i][r] - mx);
          if (vA) sacc += expf(xA - mx);
          if (vB) sacc += expf(xB - mx);
          rs[i][r] = sacc; rm[i][r] = mx;
        }
      }
  }
  // one reduction over the 32 lanes that share a row (xor < 32 stays inside the half-wave)
  const int nsplit2 = gridDim.x * 2;
  const int slot = blockIdx.x * 2 + (wave & 1);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float m = rm[i][r], s = rs[i][r];
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) {
        const float m2 = __shfl_xor(m, o, 64), s2 = __shfl_xor(s, o, 64);
        lse_merge(m, s, m2, s2);
      }
      const int row = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
      if (l31 == 0 && row < p.N) {
        p.pm[(long)slot * p.N + row] = m;
        p.ps[(long)slot * p.N + row] = s;
      }
    }
  (void)nsplit2;
}

__global__ __launch_bounds__(256) void ce_combine_kernel(int N, int nslots, const float* __restrict__ pm,
                                                         const float* __restrict__ ps,
                                                         const float* __restrict__ tgt_logit,
                                                         const int64_t* __restrict__ tgt, int64_t pad,
                                                         float* __restrict__ loss, float* __restrict__ lse) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  float m = -INFINITY, s = 0.f;
  for (int k = 0; k < nslots; ++k) lse_merge(m, s, pm[(long)k * N + n], ps[(long)k * N + n]);
  const float l = m + logf(s);
  lse[n] = l;
  loss[n] = (tgt[n] != pad) ? (l - tgt_logit[n]) : 0.f;
}

// number of vocab splits for a given problem (also sizes the workspace)
static int ce_splits(int N, int V) {
  const int rowblocks = cdiv(N, CE_BM), ntiles = cdiv(V, CE_BN);
  int want = cdiv(768, rowblocks);
  if (want > ntiles) want = ntiles;
  if (want < 1) want = 1;
  const int tps = cdiv(ntiles, want);
  return cdiv(ntiles, tps);
}

long projce_workspace_floats(int N, int V) { return (long)N * (2L * 2 * ce_splits(N, V) + 1); }

int launch_projce_forward(const float* X, const float* W, const float* bias, const int64_t* tgt, int64_t pad,
                          int N, int E, int V, float* loss, float* lse, float* ws, hipStream_t s) {
  CVAE_REQUIRE(N >= 0 && E > 0 && V > 0, "projce_forward: bad shape N=%d E=%d V=%d", N, E, V);
  if (N == 0) return 0;
  CVAE_REQUIRE(X && W && bias && tgt && loss && lse && ws, "projce_forward: null operand");
  const int nsplit = ce_splits(N, V);
  CeFwdArgs a;
  a.X = X; a.ldx = E; a.W = W; a.ldw = E; a.bias = bias; a.tgt = tgt;
  a.N = N; a.E = E; a.V = V;
  a.tiles_per_split = cdiv(cdiv(V, CE_BN), nsplit);
  a.pm = ws; a.ps = ws + (long)2 * nsplit * N; a.tgt_logit = ws + (long)4 * nsplit * N;
  a.vecX = aligned16(X) && (E % 4 == 0);
  a.vecW = aligned16(W) && (E % 4 == 0);
  // targets outside [0,V) never match a column: give them a defined logit
  CVAE_TRY(check_hip(hipMemsetAsync(a.tgt_logit, 0, (size_t)N * 4, s), "projce tgt memset"));
  {
    ProfScope prof(PK_CE_FWD, s, 2.0 * N * E * V);
    hipLaunchKernelGGL(ce_fwd_kernel, dim3(nsplit, cdiv(N, CE_BM)), dim3(256), 0, s, a);
  }
  CVAE_LAUNCH_CHECK("ce_fwd_kernel");
  hipLaunchKernelGGL(ce_combine_kernel, dim3(cdiv(N, 256)), dim3(256), 0, s, N, 2 * nsplit, a.pm, a.ps,
                     a.tgt_logit, tgt, pad, loss, lse);
  CVAE_LAUNCH_CHECK("ce_combine_kernel");
  return 0;
}

// coef[n] = g[n] * (target != pad)
__global__ void ce_coef_kernel(int N, const float* __restrict__ g, const int64_t* __restrict__ tgt, int64_t pad,
                               float* __restrict__ coef) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n < N) coef[n] = (tgt[n] != pad) ? g[n] : 0.f;
}

// out[n] += sum_m P[m, n]   (atomic accumulate; grid.y splits the rows)
__global__ __launch_bounds__(256) void colsum_kernel(int M, int N, const float* __restrict__ P, long ld,
                                                     float* __restrict__ out, int rows_per_block) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const int mbeg = blockIdx.y * rows_per_block, mend = min(M, mbeg + rows_per_block);
  float acc = 0.f;
  for (int m = mbeg; m < mend; ++m) acc += P[(long)m * ld + n];
  atomicAdd(out + n, acc);
}

int launch_colsum_acc(const float* P, int M, int N, long ld, float* out, hipStream_t s) {
  if (M == 0 || N == 0) return 0;
  int rpb = 64;
  if ((long)cdiv(N, 256) * cdiv(M, rpb) > 4096) rpb = cdiv(M, cdiv(4096, cdiv(N, 256)));
  if (deterministic_enabled()) rpb = M;          // one block per column group: one atomic per output, no order to vary
  hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(N, 256), cdiv(M, rpb)), dim3(256), 0, s, M, N, P, ld, out, rpb);
  CVAE_LAUNCH_CHECK("colsum_kernel");
  return 0;
}

long projce_bwd_chunk_rows(int N, int V) {
  const long ldp = (V + 3) / 4 * 4;
  long ch = (96L << 20) / (ldp * 4);       // keep a chunk of P within ~96 MiB (Infinity Cache is 256 MiB)
  ch = ch / 128 * 128;
  if (ch < 128) ch = 128;
  if (ch > N) ch = N;
  return ch;
}

long projce_bwd_workspace_floats(int N, int V) {
  const long ldp = (V + 3) / 4 * 4;
  return projce_bwd_chunk_rows(N, V) * ldp + N;
}

int launch_projce_backward(const float* X, const float* W, const float* bias, const int64_t* tgt, int64_t pad,
                           const float* lse, const float* gloss, int N, int E, int V, float* dX, float* dW,
                           float* db, float* ws, hipStream_t s) {
  // dX [N,E] is overwritten; dW [V,E] and db [V] are ACCUMULATED into.
  CVAE_REQUIRE(N >= 0 && E > 0 && V > 0, "projce_backward: bad shape");
  if (N == 0) return 0;
  CVAE_REQUIRE(X && W && bias && tgt && lse && gloss && dX && dW && db && ws, "projce_backward: null operand");
  const long ldp = (V + 3) / 4 * 4;
  const long CH = projce_bwd_chunk_rows(N, V);
  float* P = ws;
  float* coef = ws + CH * ldp;
  hipLaunchKernelGGL(ce_coef_kernel, dim3(cdiv(N, 256)), dim3(256), 0, s, N, gloss, tgt, pad, coef);
  CVAE_LAUNCH_CHECK("ce_coef_kernel");
  for (long r0 = 0; r0 < N; r0 += CH) {
    const int rows = (int)((N - r0) < CH ? (N - r0) : CH);
    GemmDesc g = {};
    g.transA = 0; g.transB = 1; g.M = rows; g.N = V; g.K = E;
    g.A = X + r0 * E; g.lda = E; g.B = W; g.ldb = E; g.C = P; g.ldc = ldp; g.bias = bias;
    g.act = ACT_CEGRAD; g.row_lse = lse + r0; g.row_coef = coef + r0; g.row_tgt = tgt + r0;
    CVAE_TRY(launch_gemm(g, s));
    GemmDesc dx = {};
    dx.transA = 0; dx.transB = 0; dx.M = rows; dx.N = E; dx.K = V;
    dx.A = P; dx.lda = ldp; dx.B = W; dx.ldb = E; dx.C = dX + r0 * E; dx.ldc = E;
    CVAE_TRY(launch_gemm(dx, s));
    GemmDesc dw = {};
    dw.transA = 1; dw.transB = 0; dw.M = V; dw.N = E; dw.K = rows;
    dw.A = P; dw.lda = ldp; dw.B = X + r0 * E; dw.ldb = E; dw.C = dW; dw.ldc = E; dw.beta = 1;
    CVAE_TRY(launch_gemm(dw, s));
    CVAE_TRY(launch_colsum_acc(P, rows, V, ldp, db, s));
  }
  return 0;
}

}  // namespace cvae
