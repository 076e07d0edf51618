// Hidden->vocab projection + softmax cross-entropy for E <= 200 (every shipped config has
// emb_dim = 200): the hot kernels of both the training step and the IW-NLL evaluation.
// Reference: ToVocab + SeqLoss, CV/Modules/vae_couple.py:119, CV/Modules/Losses/SeqLoss.py:36-44.
//
// What is different from a dense (N,V) logits GEMM:
//  * Padding rows (target == pad) contribute neither loss nor gradient.  A scan builds the
//    compact list of live rows on the device (no host sync); every kernel below works on that
//    list and reads the live count from device memory, so ~half the rows of a Yelp batch are
//    never multiplied at all.
//  * ce_core_kernel is persistent (2 blocks per CU).  A block takes (128 live rows) x (a range
//    of 32-wide vocab tiles) work items; the split of the vocabulary is chosen on the device
//    from the live count so that the items fill the chip.  The 128 rows stay in REGISTERS for
//    the whole item (each lane holds its row's K values, laid out as v_mfma_f32_32x32x2_f32
//    B-fragments); only W tiles stream through LDS, double buffered, one barrier per tile.
//  * The product is formed transposed, S^T = W_tile . X^T: a lane then owns one ROW and 16 of
//    the tile's 32 vocab columns, so the online softmax state is two registers per lane and the
//    gradient tile leaves the accumulators as four 16-byte stores per lane.
//  * forward  (GRAD = false): per-(row, split) running (max, sum-exp) + target logit; a
//    combine kernel merges the splits and scatters loss / lse to the original rows.
//  * backward (GRAD = true): the same core writes P = coef * (softmax - onehot) for a chunk of
//    live rows; gemm224_kernel then forms dX = P W (rows scattered back through the list) and
//    [dW | db] += P^T [X | 1] (X rows gathered through the list) with exact-fp32 MFMA, split-K.
#include "common.hpp"
#include "misc.hpp"

namespace cvae {

constexpr int CE_ROWS = 128;   // live rows per work item (4 waves x 32)
constexpr int CE_TN = 32;      // vocab columns per tile
constexpr int CE_LDW = 204;    // LDS row stride of a W tile: 200 k-values, bias at column 200; 204/4 odd -> b128 reads conflict-free
constexpr int CE_BIAS_COL = 200;

// ------------------------------------------------------------------------------------------
// live-row list
// per-row outputs that start at zero (dead rows: loss, lse; every row: tl), for every decoder that shares the list
struct CeRowsOut { static constexpr int MAX = 4; int n; float* loss[MAX]; float* lse[MAX]; float* tl[MAX]; };
__global__ __launch_bounds__(1024) void ce_rows_count_kernel(int N, int TR, const int64_t* __restrict__ tgt,
                                                             int64_t pad, int* __restrict__ counts, CeRowsOut o) {
  __shared__ int wc[16];
  const int n = blockIdx.x * 1024 + threadIdx.x;
  const bool live = n < N && tgt[n % TR] != pad;
  if (n < N) {
    for (int i = 0; i < o.n; ++i) {
      if (!live) {
        if (o.loss[i]) o.loss[i][n] = 0.f;
        if (o.lse[i]) o.lse[i][n] = 0.f;
      }
      if (o.tl[i]) o.tl[i][n] = 0.f;
    }
  }
  const unsigned long long b = __ballot(live);
  if ((threadIdx.x & 63) == 0) wc[threadIdx.x >> 6] = __popcll(b);
  __syncthreads();
  if (threadIdx.x == 0) {
    int c = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) c += wc[i];
    counts[blockIdx.x] = c;
  }
}

__global__ __launch_bounds__(1024) void ce_rows_write_kernel(int N, int TR, const int64_t* __restrict__ tgt,
                                                             int64_t pad, const int* __restrict__ counts,
                                                             int* __restrict__ rows, int* __restrict__ nlive) {
  __shared__ int red[16];
  __shared__ int wc[16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int part = 0;
  for (int i = threadIdx.x; i < (int)blockIdx.x; i += 1024) part += counts[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
  const int n = blockIdx.x * 1024 + threadIdx.x;
  const bool live = n < N && tgt[n % TR] != pad;
  const unsigned long long b = __ballot(live);
  if (lane == 0) {
    red[wave] = part;
    wc[wave] = __popcll(b);
  }
  __syncthreads();
  int off = 0, woff = 0, total = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    off += red[i];
    if (i < wave) woff += wc[i];
    total += wc[i];
  }
  if (live) rows[off + woff + __popcll(b & ((1ull << lane) - 1ull))] = n;
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *nlive = off + total;
}

// ------------------------------------------------------------------------------------------
// vocabulary split chosen on the device (forward, combine and backward must agree on it)
__device__ __forceinline__ int ce_pick_split(int rb, int ntiles, int cap, int nblk) {
  int smax = cap < ntiles ? cap : ntiles;
  if (smax < 1) smax = 1;
  int best = 1;
  float bc = 3.0e38f;
  for (int s = 1; s <= smax; ++s) {
    const int items = rb * s;
    const int rounds = (items + nblk - 1) / nblk;
    const float c = (float)rounds / (float)s + 0.002f * (float)s;   // small price per extra split (partials, X reloads)
    if (c < bc) { bc = c; best = s; }
  }
  const int tps = (ntiles + best - 1) / best;
  return (ntiles + tps - 1) / tps;        // no empty splits
}

struct CeArgs {
  const float* X;        // [N,E] rows of the decoder output
  const float* W;        // [V,E]
  const float* bias;     // [V]
  const int64_t* tgt;    // [TR] targets, shared by the N/TR segments
  const int* rows;       // compact list of live rows
  const int* nlive;      // device count
  int N, TR, E, V, vec;
  int cap;               // forward: splits the partial workspace holds
  long pstride;          // forward: floats between split slots
  float* pm; float* ps; float* tl;
  // GRAD
  const float* lse;      // [N]
  const float* gsent;    // [(N/TR) * R]
  int R;
  float* P; long ldp;    // [rcount, ldp] chunk of the gradient w.r.t. the logits (live rows r0 ...)
  int r0, rcount;
};

__device__ __forceinline__ float4 load4_guard(const float* __restrict__ src, int k, int K, int vec) {
  float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
  if (vec && k + 3 < K) {
    r = *reinterpret_cast<const float4*>(src);
  } else {
    if (k + 0 < K) r.x = src[0];
    if (k + 1 < K) r.y = src[1];
    if (k + 2 < K) r.z = src[2];
    if (k + 3 < K) r.w = src[3];
  }
  return r;
}

__device__ __forceinline__ void lse_merge2(float& m, float& s, float m2, float s2) {
  const float M = fmaxf(m, m2);
  const float a = (m == M) ? 1.f : expf(m - M);     // (m == M) also covers -inf == -inf
  const float b = (m2 == M) ? 1.f : expf(m2 - M);
  s = s * a + s2 * b;
  m = M;
}

// NQ: K is covered by NQ groups of 8 (zero padded): E <= 8*NQ <= 200.
// SAVE (forward of a training step): also store the logits of the live rows, S[live index][v], so that the
// backward pass turns them into the gradient in place (ce_grad_from_logits_kernel) instead of recomputing
// the product.  The stores ride along with an MFMA-bound kernel.
template <int NQ, bool GRAD, bool EXACT, bool SAVE = false>
__global__ __launch_bounds__(256, 2) void ce_core_kernel(CeArgs p) {
  __shared__ __attribute__((aligned(16))) float Wt[2][CE_TN * CE_LDW];
  constexpr int F4ROW = 2 * NQ;                 // float4 per W row
  constexpr int NF4 = CE_TN * F4ROW;            // float4 per tile
  constexpr int NST = (NF4 + 255) / 256;        // staged float4 per thread
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;

  const int nlive = *p.nlive;
  int rbeg = 0, rend = nlive;
  if (GRAD) {
    rbeg = p.r0;
    rend = min(nlive, p.r0 + p.rcount);
  }
  if (rend <= rbeg) return;
  const int rb = (rend - rbeg + CE_ROWS - 1) / CE_ROWS;
  const int ntiles = (p.V + CE_TN - 1) / CE_TN;
  const int nsplit = ce_pick_split(rb, ntiles, p.cap, gridDim.x);
  const int tps = (ntiles + nsplit - 1) / nsplit;

  // exact K and 16-byte rows: no per-element guards anywhere (uniform branch)
  constexpr bool exactk = EXACT;     // host guarantees: 16-byte aligned operands and E == 8 * NQ
  float4 st[NST];
  float bst = 0.f;
  auto stage_load = [&](int tile) {
    const int n0 = tile * CE_TN;
    if constexpr (exactk) {
      // the tile is one contiguous block of W: float4 number idx of it
      const float4* src = reinterpret_cast<const float4*>(p.W + (long)n0 * p.E);
      const int lim = min(NF4, (p.V - n0) * F4ROW);
#pragma unroll
      for (int i = 0; i < NST; ++i) {
        const int idx = tid + 256 * i;
        st[i] = (idx < lim) ? src[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    } else {
#pragma unroll
      for (int i = 0; i < NST; ++i) {
        const int idx = tid + 256 * i;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (idx < NF4) {
          const int row = idx / F4ROW, k = 4 * (idx % F4ROW);
          const int vr = n0 + row;
          if (vr < p.V) v = load4_guard(p.W + (long)vr * p.E + k, k, p.E, p.vec);
        }
        st[i] = v;
      }
    }
    if (tid < CE_TN) bst = (n0 + tid < p.V) ? p.bias[n0 + tid] : -INFINITY;
  };
  auto stage_store = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NST; ++i) {
      const int idx = tid + 256 * i;
      if (idx < NF4) {
        const int row = idx / F4ROW, k = 4 * (idx % F4ROW);
        *reinterpret_cast<float4*>(&Wt[buf][row * CE_LDW + k]) = st[i];
      }
    }
    if (tid < CE_TN) Wt[buf][tid * CE_LDW + CE_BIAS_COL] = bst;
  };

  for (int item = blockIdx.x; item < rb * nsplit; item += gridDim.x) {
    const int sp = item % nsplit, rbi = item / nsplit;
    const int t0 = sp * tps, t1 = min(ntiles, t0 + tps);
    const int ci = rbeg + rbi * CE_ROWS + wave * 32 + l31;     // index into the live list
    const bool rv = ci < rend;
    const int n = rv ? p.rows[ci] : 0;
    // this lane's row as MFMA B-fragments: step (q, s) uses k = 8q + 4*lhi + s
    float x[4 * NQ];
    if constexpr (exactk) {
      const float4* xr = reinterpret_cast<const float4*>(p.X + (long)n * p.E) + lhi;
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const float4 v = rv ? xr[2 * q] : make_float4(0.f, 0.f, 0.f, 0.f);
        x[4 * q + 0] = v.x; x[4 * q + 1] = v.y; x[4 * q + 2] = v.z; x[4 * q + 3] = v.w;
      }
    } else {
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int k = 8 * q + 4 * lhi;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (rv) v = load4_guard(p.X + (long)n * p.E + k, k, p.E, p.vec);
        x[4 * q + 0] = v.x; x[4 * q + 1] = v.y; x[4 * q + 2] = v.z; x[4 * q + 3] = v.w;
      }
    }
    int tg = -1;
    if (rv) {
      const int64_t t = p.tgt[n % p.TR];
      if (t >= 0 && t < p.V) tg = (int)t;
    }
    float m = -INFINITY, s = 0.f, tlv = 0.f;
    bool found = false;
    float lse_r = 0.f, coef = 0.f;
    float* prow = nullptr;
    if (GRAD) {
      if (rv) {
        lse_r = p.lse[n];
        coef = p.gsent[(n / p.TR) * p.R + n % p.R];
      }
      prow = p.P + (long)(ci - p.r0) * p.ldp;
    }
    if (SAVE) prow = p.P + (long)ci * p.ldp;

    stage_load(t0);
    stage_store(0);
    __syncthreads();
    for (int t = t0; t < t1; ++t) {
      const int buf = (t - t0) & 1;
      const int n0 = t * CE_TN;
      if (t + 1 < t1) stage_load(t + 1);
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const float* wl = &Wt[buf][l31 * CE_LDW + 4 * lhi];
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const float4 w4 = *reinterpret_cast<const float4*>(wl + 8 * q);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.x, x[4 * q + 0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.y, x[4 * q + 1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.z, x[4 * q + 2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.w, x[4 * q + 3], acc, 0, 0, 0);
      }
      // acc[r] = logit(row = this lane's, vocab column n0 + j), j = (r & 3) + 8 * (r >> 2) + 4 * lhi
      float xv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int j = (r & 3) + 8 * (r >> 2) + 4 * lhi;
        xv[r] = acc[r] + Wt[buf][j * CE_LDW + CE_BIAS_COL];     // -inf beyond V
      }
      if (!GRAD) {
        float mx = xv[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, xv[r]);
        const float mn = fmaxf(m, mx);
        const float mref = (mn == -INFINITY) ? 0.f : mn;
        float sum = s * __expf(m - mref);
#pragma unroll
        for (int r = 0; r < 16; ++r) sum += __expf(xv[r] - mref);
        s = sum;
        m = mn;
        const int jt = tg - n0 - 4 * lhi;
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (jt == (r & 3) + 8 * (r >> 2)) { tlv = xv[r]; found = true; }
        if (SAVE && rv) {
#pragma unroll
          for (int g = 0; g < 4; ++g)
            *reinterpret_cast<float4*>(prow + n0 + 8 * g + 4 * lhi) =
                make_float4(xv[4 * g], xv[4 * g + 1], xv[4 * g + 2], xv[4 * g + 3]);
        }
      } else {
        const int jt = tg - n0 - 4 * lhi;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float o[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int r = 4 * g + i;
            o[i] = coef * (__expf(xv[r] - lse_r) - ((jt == i + 8 * g) ? 1.f : 0.f));
          }
          if (rv) *reinterpret_cast<float4*>(prow + n0 + 8 * g + 4 * lhi) = make_float4(o[0], o[1], o[2], o[3]);
        }
      }
      if (t + 1 < t1) stage_store(buf ^ 1);
      __syncthreads();
    }
    if (!GRAD) {
      const float m2 = __shfl_xor(m, 32, 64), s2 = __shfl_xor(s, 32, 64);
      lse_merge2(m, s, m2, s2);
      if (rv) {
        if (lhi == 0) {
          p.pm[(long)sp * p.pstride + ci] = m;
          p.ps[(long)sp * p.pstride + ci] = s;
        }
        if (found) p.tl[ci] = tlv;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// The same core with the product on the bf16 matrix pipe, error-compensated ("bf16x3"): every fp32 operand is
// split into hi = bf16(v) and lo = bf16(v - hi), and  x.w  is accumulated in fp32 as  hi.hi + hi.lo + lo.hi
// (v_mfma_f32_32x32x16_bf16: 16x the rate of the fp32-input MFMA, so 3 passes cost 3/16 of it).  The dropped
// lo.lo term and the 16-bit truncation of hi + lo leave a relative error of ~2^-16 per product; for a logit (a sum
// of E = 200 products of magnitude ~0.03) that is an absolute error of a few 1e-6, measured against fp64 in
// tests/test_gpu_kernels.py::test_projce_bf16x3_error (the fp32-input kernel above: ~1e-7; the bars: 1e-4 / 1e-3).
// W is split ONCE per launch into the tile-blocked image the LDS wants (ce_presplit_kernel: per 32-row tile a
// [32][LDR] block of hi then one of lo, LDR = 16*NB + 8 bf16 -> row stride an odd number of 16-byte units: the
// b128 fragment reads are conflict-free), so staging is a straight 16-byte copy; a row's K values are split in
// registers when the item starts.  Everything else -- live-row list, persistent items, vocabulary split, S^T
// formulation, online softmax, SAVE / GRAD epilogues -- is the fp32 kernel's.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void bf3_split(f32x8 v, bf16x8& hi, bf16x8& lo) {
  hi = __builtin_convertvector(v, bf16x8);
  lo = __builtin_convertvector(v - __builtin_convertvector(hi, f32x8), bf16x8);
}

// scale / bias (evaluation launch, ce_core_bf3_kernel<.., FAST>): every entry times `scale` (log2 e: the kernel's
// softmax then runs on v_exp_f32 without a multiply), and column k = E of the image -- a K slot the 16-wide blocks
// leave free when E is not a multiple of 16 -- holds scale * bias[v] (the rows carry 1.0 there: the bias comes out of
// the product), -1e30 for the rows past V (exp2 of it is 0: no masking in the kernel)
template <int NB>
__global__ __launch_bounds__(256) void ce_presplit_kernel(int V, int E, const float* __restrict__ W, int vec,
                                                          bf16x8* __restrict__ Wp, float scale = 1.f,
                                                          const float* __restrict__ bias = nullptr) {
  constexpr int LDR8 = 2 * NB + 1;                       // bf16x8 units per LDS row
  const int ntiles = (V + CE_TN - 1) / CE_TN;
  const long total = (long)ntiles * CE_TN * LDR8;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int u = (int)(i % LDR8), row = (int)((i / LDR8) % CE_TN), tile = (int)(i / ((long)LDR8 * CE_TN));
    const int v = tile * CE_TN + row, k = 8 * u;
    f32x8 x;
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = 0.f;
    if (v < V && k < E) {
      const float* src = W + (long)v * E + k;
      if (vec && k + 7 < E) {
        const float4 a = *reinterpret_cast<const float4*>(src), b = *reinterpret_cast<const float4*>(src + 4);
        x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w; x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w;
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) if (k + j < E) x[j] = src[j];
      }
    }
    if (bias) {
#pragma unroll
      for (int j = 0; j < 8; ++j) x[j] *= scale;
      if (E >= k && E < k + 8) {
        const float bv = v < V ? scale * bias[v] : -1e30f;
#pragma unroll
        for (int j = 0; j < 8; ++j) if (k + j == E) x[j] = bv;
      }
    }
    bf16x8 hi, lo;
    bf3_split(x, hi, lo);
    bf16x8* t = Wp + (long)tile * 2 * CE_TN * LDR8;
    t[row * LDR8 + u] = hi;
    t[CE_TN * LDR8 + row * LDR8 + u] = lo;
  }
}

// FAST (the evaluation launch: no GRAD, no SAVE, E < 16 NB): W's image is pre-scaled by log2 e and carries the bias
// in K slot E (ce_presplit_kernel), so a tile's epilogue is max / subtract / v_exp_f32 / add per logit plus the
// target pick -- no bias add, no -inf masking, no multiply (183 -> ~125 vector instructions per tile and wave).
// (m, target logit) go back to natural units when they are stored.  Worth 2-6 % of the launch: the kernel runs at
// 1.81-1.87 GHz (in-kernel clock, CE_EXP_STAMPS) with the matrix pipe ~65 % busy, and cycles saved come back partly
// as clock.
#ifdef CE_EXP_STAMPS
#define CE_STAMP(k) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); sacc_[k] += (unsigned)(now_ - tprev_); tprev_ = now_; } while (0)
#else
#define CE_STAMP(k) do {} while (0)
#endif
template <int NB, bool GRAD, bool SAVE, bool FAST = false>
__global__ __launch_bounds__(256, 2) void ce_core_bf3_kernel(CeArgs p, const bf16x8* __restrict__ Wp) {
  static_assert(!FAST || (!GRAD && !SAVE), "FAST is the evaluation form");
#ifdef CE_EXP_STAMPS
  unsigned long long sacc_[5] = {0, 0, 0, 0, 0};
  unsigned long long tprev_ = __builtin_amdgcn_s_memtime();
  const unsigned long long tbeg_ = tprev_, rbeg_ = __builtin_amdgcn_s_memrealtime();
  unsigned ntl_ = 0;
#endif
  constexpr int LDR8 = 2 * NB + 1;
  constexpr int TILE8 = 2 * CE_TN * LDR8;        // bf16x8 units per tile (hi block + lo block)
  constexpr int NST = (TILE8 + 255) / 256;
  __shared__ __attribute__((aligned(16))) bf16x8 Wt[2][TILE8];
  __shared__ float Bt[2][CE_TN];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;

  const int nlive = *p.nlive;
  int rbeg = 0, rend = nlive;
  if (GRAD) {
    rbeg = p.r0;
    rend = min(nlive, p.r0 + p.rcount);
  }
  if (rend <= rbeg) return;
  const int rb = (rend - rbeg + CE_ROWS - 1) / CE_ROWS;
  const int ntiles = (p.V + CE_TN - 1) / CE_TN;
  const int nsplit = ce_pick_split(rb, ntiles, p.cap, gridDim.x);
  const int tps = (ntiles + nsplit - 1) / nsplit;

  // A tile of W's image (already in LDS order: ce_presplit_kernel) is copied global -> LDS by the LDS-DMA, one 1 KB
  // piece per wave-instruction, issued when the previous tile's barrier has released the other buffer; it lands
  // while this tile is multiplied (hipcc drains it at the next barrier).  Staging through registers cost every wave
  // 7 ds_write_b128 per tile: 55 KB per CU and tile pair through the LDS store path (~79 B/clk) = ~700 cycles in
  // which no wave of the CU issued an MFMA (both blocks of a CU run in phase).
  static_assert(TILE8 % 64 == 0, "whole DMA pieces");
  constexpr int NPC = TILE8 / 64;
  (void)NST;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  float bst = 0.f;
  auto stage_load = [&](int tile, int buf) {
    const bf16x8* src = Wp + (long)tile * TILE8 + lane;
#pragma unroll
    for (int j = 0; j < (NPC + 3) / 4; ++j) {
      const int pc = wave_u + 4 * j;
      if (pc < NPC)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + pc * 64),
                                         (__attribute__((address_space(3))) void*)(&Wt[buf][pc * 64]), 16, 0, 0);
    }
    const int n0 = tile * CE_TN;
    if (!FAST && tid < CE_TN) bst = (n0 + tid < p.V) ? p.bias[n0 + tid] : -INFINITY;
  };
  auto stage_store = [&](int buf) {
    if (!FAST && tid < CE_TN) Bt[buf][tid] = bst;
  };


  for (int item = blockIdx.x; item < rb * nsplit; item += gridDim.x) {
    const int sp = item % nsplit, rbi = item / nsplit;
    const int t0 = sp * tps, t1 = min(ntiles, t0 + tps);
    const int ci = rbeg + rbi * CE_ROWS + wave * 32 + l31;     // index into the live list
    const bool rv = ci < rend;
    const int n = rv ? p.rows[ci] : 0;
    // this lane's row as B fragments of the 32x32x16 MFMA: block Q uses k = 16Q + 8*lhi + (0..7)
    bf16x8 xh[NB], xl[NB];
    {
      const float* xr = p.X + (long)n * p.E + 8 * lhi;
#pragma unroll
      for (int Q = 0; Q < NB; ++Q) {
        f32x8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = 0.f;
        if (rv && 16 * Q + 8 * lhi < p.E) {       // E % 8 == 0 and 16-byte rows (host-checked): whole groups of 8
          const float4 a = *reinterpret_cast<const float4*>(xr + 16 * Q), b = *reinterpret_cast<const float4*>(xr + 16 * Q + 4);
          v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        }
        if (FAST) {                                 // the bias slot: 1.0 at k = E
#pragma unroll
          for (int j = 0; j < 8; ++j) if (16 * Q + 8 * lhi + j == p.E) v[j] = 1.f;
        }
        bf3_split(v, xh[Q], xl[Q]);
      }
    }
    int tg = -1;
    if (rv) {
      const int64_t t = p.tgt[n % p.TR];
      if (t >= 0 && t < p.V) tg = (int)t;
    }
    float m = -INFINITY, s = 0.f, tlv = 0.f;
    bool found = false;
    float lse_r = 0.f, coef = 0.f;
    float* prow = nullptr;
    if (GRAD) {
      if (rv) {
        lse_r = p.lse[n];
        coef = p.gsent[(n / p.TR) * p.R + n % p.R];
      }
      prow = p.P + (long)(ci - p.r0) * p.ldp;
    }
    if (SAVE) prow = p.P + (long)ci * p.ldp;

    stage_load(t0, 0);
    stage_store(0);
    __syncthreads();
    for (int t = t0; t < t1; ++t) {
      const int buf = (t - t0) & 1;
      const int n0 = t * CE_TN;
      CE_STAMP(3);
#ifndef CE_EXP_NOLOAD
      if (t + 1 < t1) stage_load(t + 1, buf ^ 1);
#endif
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const bf16x8* wh = &Wt[buf][l31 * LDR8 + lhi];
      const bf16x8* wl = wh + CE_TN * LDR8;
      // fragments are fetched CE_PF K blocks ahead of their MFMAs (register ring): an LDS round trip is longer
      // than the three MFMAs of one block, and the scheduler otherwise sinks each read next to its use
#ifndef CE_PF
#define CE_PF 3
#endif
      bf16x8 fa[CE_PF + 1], fl[CE_PF + 1];
#pragma unroll
      for (int Q = 0; Q < CE_PF && Q < NB; ++Q) { fa[Q] = wh[2 * Q]; fl[Q] = wl[2 * Q]; }
#pragma unroll
      for (int Q = 0; Q < NB; ++Q) {
        if (Q + CE_PF < NB) { fa[(Q + CE_PF) % (CE_PF + 1)] = wh[2 * (Q + CE_PF)]; fl[(Q + CE_PF) % (CE_PF + 1)] = wl[2 * (Q + CE_PF)]; }
        __builtin_amdgcn_sched_barrier(0);
#ifdef CE_EXP_NOLDS
        const bf16x8 ah = xh[(Q + 1) % NB], al = xl[(Q + 1) % NB];
#else
        const bf16x8 ah = fa[Q % (CE_PF + 1)], al = fl[Q % (CE_PF + 1)];
#endif
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, xl[Q], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, xh[Q], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, xh[Q], acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      CE_STAMP(0);
      // acc[r] = logit(row = this lane's, vocab column n0 + j), j = (r & 3) + 8 * (r >> 2) + 4 * lhi
      if constexpr (FAST) {                       // base-2 logits, bias included, columns past V at -1e30
        // (asm: fmaxf on MFMA results costs a canonicalising v_max per operand.  hipcc pads no hazard inside an asm
        // statement: the wait states from the last MFMA to the first reader of its result open the string)
        float mx;
        asm volatile("s_nop 15\n\ts_nop 3\n\t"
                     "v_max3_f32 %0, %1, %2, %3\n\t"
                     "v_max3_f32 %0, %0, %4, %5\n\t"
                     "v_max3_f32 %0, %0, %6, %7\n\t"
                     "v_max3_f32 %0, %0, %8, %9\n\t"
                     "v_max3_f32 %0, %0, %10, %11\n\t"
                     "v_max3_f32 %0, %0, %12, %13\n\t"
                     "v_max3_f32 %0, %0, %14, %15\n\t"
                     "v_max3_f32 %0, %0, %16, %17"
                     : "=&v"(mx)
                     : "v"(m), "v"(acc[0]), "v"(acc[1]), "v"(acc[2]), "v"(acc[3]), "v"(acc[4]), "v"(acc[5]), "v"(acc[6]),
                       "v"(acc[7]), "v"(acc[8]), "v"(acc[9]), "v"(acc[10]), "v"(acc[11]), "v"(acc[12]), "v"(acc[13]),
                       "v"(acc[14]), "v"(acc[15]));
        const float mn = mx;
        float sum = s * __builtin_amdgcn_exp2f(m - mn);                               // m = -inf at first: 0
#pragma unroll
        for (int r = 0; r < 16; ++r) sum += __builtin_amdgcn_exp2f(acc[r] - mn);
        s = sum;
        m = mn;
        // (no "is any row's target in this tile" branch around this: it saved the 34 instructions on 19 tiles of 20, but
        // the blocks of an XCD then drifted apart along the vocabulary and W's tiles stopped being shared in L2 --
        // 42 GB instead of 9 GB fetched from the fabric per launch, at the same speed)
        {
          const int jt = tg - n0 - 4 * lhi;
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (jt == (r & 3) + 8 * (r >> 2)) { tlv = acc[r]; found = true; }
        }
        CE_STAMP(1);
        if (t + 1 < t1) stage_store(buf ^ 1);
        CE_STAMP(2);
        __syncthreads();
#ifdef CE_EXP_STAMPS
        ++ntl_;
#endif
        continue;
      }
      float xv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int j = (r & 3) + 8 * (r >> 2) + 4 * lhi;
        xv[r] = acc[r] + Bt[buf][j];     // -inf beyond V
      }
#ifdef CE_EXP_NOEPI
      if (!GRAD && !SAVE) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s += xv[r];
      } else
#endif
      if (!GRAD) {
        float mx = xv[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, xv[r]);
        const float mn = fmaxf(m, mx);
        const float mref = (mn == -INFINITY) ? 0.f : mn;
        float sum = s * __expf(m - mref);
#pragma unroll
        for (int r = 0; r < 16; ++r) sum += __expf(xv[r] - mref);
        s = sum;
        m = mn;
        const int jt = tg - n0 - 4 * lhi;
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (jt == (r & 3) + 8 * (r >> 2)) { tlv = xv[r]; found = true; }
        if (SAVE && rv) {
#pragma unroll
          for (int g = 0; g < 4; ++g)
            *reinterpret_cast<float4*>(prow + n0 + 8 * g + 4 * lhi) =
                make_float4(xv[4 * g], xv[4 * g + 1], xv[4 * g + 2], xv[4 * g + 3]);
        }
      } else {
        const int jt = tg - n0 - 4 * lhi;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float o[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int r = 4 * g + i;
            o[i] = coef * (__expf(xv[r] - lse_r) - ((jt == i + 8 * g) ? 1.f : 0.f));
          }
          if (rv) *reinterpret_cast<float4*>(prow + n0 + 8 * g + 4 * lhi) = make_float4(o[0], o[1], o[2], o[3]);
        }
      }
      if (t + 1 < t1) stage_store(buf ^ 1);
      __syncthreads();
    }
    CE_STAMP(4);
    if (!GRAD) {
      if (FAST) {                                 // back to natural units: s is a sum of e^(x - m) either way
        m *= 0.6931471805599453f;
        tlv *= 0.6931471805599453f;
      }
      const float m2 = __shfl_xor(m, 32, 64), s2 = __shfl_xor(s, 32, 64);
      lse_merge2(m, s, m2, s2);
      if (rv) {
        if (lhi == 0) {
          p.pm[(long)sp * p.pstride + ci] = m;
          p.ps[(long)sp * p.pstride + ci] = s;
        }
        if (found) p.tl[ci] = tlv;
      }
    }
  }
#ifdef CE_EXP_STAMPS
  if (FAST && (blockIdx.x == 0 || blockIdx.x == 300) && (tid == 0 || tid == 192) && ntl_ > 1000)
    printf("ce stamps block %d tid %d: tiles %u, per tile: mfma %llu, epilogue %llu, stage_store %llu, barrier+top %llu; per item other %llu; clock %.0f MHz over %.2f ms\n",
           (int)blockIdx.x, tid, ntl_, sacc_[0] / ntl_, sacc_[1] / ntl_, sacc_[2] / ntl_, sacc_[3] / ntl_, sacc_[4],
           100.0 * (double)(__builtin_amdgcn_s_memtime() - tbeg_) / (double)(__builtin_amdgcn_s_memrealtime() - rbeg_),
           (double)(__builtin_amdgcn_s_memrealtime() - rbeg_) / 1e5);
#endif
}

__global__ __launch_bounds__(256) void ce_combine2_kernel(CeArgs p, int nblk, float* __restrict__ loss,
                                                          float* __restrict__ lse) {
  const int nlive = *p.nlive;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nlive) return;
  const int rb = (nlive + CE_ROWS - 1) / CE_ROWS;
  const int nsplit = ce_pick_split(rb, (p.V + CE_TN - 1) / CE_TN, p.cap, nblk);
  // two passes with independent loads (a serial merge of up to 64 partials was a 17 us chain of load + exp round trips)
  const float* pm = p.pm + i;
  const float* ps = p.ps + i;
  float m = -INFINITY, s = 0.f;
#pragma unroll 8
  for (int k = 0; k < nsplit; ++k) m = fmaxf(m, pm[(long)k * p.pstride]);
#pragma unroll 8
  for (int k = 0; k < nsplit; ++k) {
    const float m2 = pm[(long)k * p.pstride], s2 = ps[(long)k * p.pstride];
    s += (m2 == m) ? s2 : s2 * __expf(m2 - m);
  }
  const float l = m + logf(s);
  const int n = p.rows[i];
  lse[n] = l;
  loss[n] = l - p.tl[i];
}

// P[i][v] = coef * (exp(S[i][v] - lse) - (v == target)) in place on the logits the forward pass saved, for
// the live rows [0, *nlive): 2 x 4 bytes per logit of HBM traffic instead of the E-long product again.
__global__ __launch_bounds__(256) void ce_grad_from_logits_kernel(CeArgs p) {
  const int nlive = *p.nlive;
  const int V4 = p.V / 4;
  for (int ci = blockIdx.x; ci < nlive; ci += gridDim.x) {
    const int n = p.rows[ci];
    const float lse_r = p.lse[n];
    const float coef = p.gsent[(n / p.TR) * p.R + n % p.R];
    const int64_t t = p.tgt[n % p.TR];
    const int tg = (t >= 0 && t < p.V) ? (int)t : -1;
    float* row = p.P + (long)ci * p.ldp;
    float4* row4 = reinterpret_cast<float4*>(row);
    for (int i = threadIdx.x; i < V4; i += 256) {
      float4 x = row4[i];
      const int v = 4 * i;
      x.x = coef * (__expf(x.x - lse_r) - (v == tg ? 1.f : 0.f));
      x.y = coef * (__expf(x.y - lse_r) - (v + 1 == tg ? 1.f : 0.f));
      x.z = coef * (__expf(x.z - lse_r) - (v + 2 == tg ? 1.f : 0.f));
      x.w = coef * (__expf(x.w - lse_r) - (v + 3 == tg ? 1.f : 0.f));
      row4[i] = x;
    }
    for (int v = 4 * V4 + threadIdx.x; v < p.V; v += 256)
      row[v] = coef * (__expf(row[v] - lse_r) - (v == tg ? 1.f : 0.f));
  }
}

// ------------------------------------------------------------------------------------------
// host side
static int ce_cap(long N, int V) {
  const long rb = (N + CE_ROWS - 1) / CE_ROWS;
  long cap = (2L * persistent_grid_blocks() + rb - 1) / rb;
  if (cap > 64) cap = 64;
  if (cap < 1) cap = 1;
  const int ntiles = cdiv(V, CE_TN);
  return (int)(cap < ntiles ? cap : ntiles);
}

bool ce_fast_supported(int E) { return E >= 1 && E <= 200; }

// bf16x3 path: 16-byte rows, E a multiple of 8 (every shipped config: E = 200); cvae_split_bf16(0) / CVAE_SPLIT_BF16=0
// selects the fp32-input MFMA kernel instead (diagnostics, error measurements)
static int bf3_blocks(int E) {                   // 16-wide K blocks, rounded up to an instantiated count
  const int nb = (E + 15) / 16;
  return nb <= 1 ? 1 : nb <= 2 ? 2 : nb <= 4 ? 4 : nb <= 7 ? 7 : 13;
}
static bool bf3_ok(const CeArgs& a) { return split_bf16_enabled() && a.vec && a.E % 8 == 0 && a.E <= 208; }
// floats of workspace for the split image of W (16-byte aligned start included)
static long bf3_presplit_floats(int E, int V) {
  const long ntiles = cdiv(V, CE_TN);
  return ntiles * 2 * CE_TN * (2 * bf3_blocks(E) + 1) * 4 + 8;
}


// workspace (floats): rows[N] | counts | nlive | tl[N] | pm[cap*N] | ps[cap*N]
static long ce_counts_slots(long N) { return (N + 1023) / 1024 + 8; }
static long ce_fwd_base_floats(long N, int V) { return N + ce_counts_slots(N) + N + 2L * ce_cap(N, V) * N + 16; }
long ce_fast_workspace_floats(long N, int V) {
  return ce_fwd_base_floats(N, V) + bf3_presplit_floats(208, V);       // + the split image of W (bf16x3 path)
}

static int launch_rows_n(const int64_t* tgt, int64_t pad, long N, long TR, int* rows, int* counts, int* nlive,
                         const CeRowsOut& o, hipStream_t s) {
  const int nb = cdiv(N, 1024);
  hipLaunchKernelGGL(ce_rows_count_kernel, dim3(nb), dim3(1024), 0, s, (int)N, (int)TR, tgt, pad, counts, o);
  CVAE_LAUNCH_CHECK("ce_rows_count_kernel");
  hipLaunchKernelGGL(ce_rows_write_kernel, dim3(nb), dim3(1024), 0, s, (int)N, (int)TR, tgt, pad, counts, rows, nlive);
  CVAE_LAUNCH_CHECK("ce_rows_write_kernel");
  return 0;
}
static int launch_rows(const int64_t* tgt, int64_t pad, long N, long TR, int* rows, int* counts, int* nlive,
                       float* loss, float* lse, float* tl, hipStream_t s) {
  CeRowsOut o = {};
  o.n = 1; o.loss[0] = loss; o.lse[0] = lse; o.tl[0] = tl;
  return launch_rows_n(tgt, pad, N, TR, rows, counts, nlive, o, s);
}

template <bool GRAD, bool SAVE = false>
static int launch_core(const CeArgs& a, hipStream_t s, float* presplit_ws = nullptr) {
  const dim3 grid(persistent_grid_blocks()), block(256);
  if (presplit_ws && bf3_ok(a)) {
    bf16x8* Wp = reinterpret_cast<bf16x8*>((reinterpret_cast<uintptr_t>(presplit_ws) + 15) & ~(uintptr_t)15);
    const int nb = bf3_blocks(a.E);
    const long units = (long)cdiv(a.V, CE_TN) * CE_TN * (2 * nb + 1);
    int pg = (int)((units + 255) / 256);
    if (pg > 4096) pg = 4096;
    // the evaluation form (FAST) needs a free K slot for the bias: E < 16 nb; env CVAE_CE_FAST_EVAL=0 turns it off
    static int fast_on = -1;
    if (fast_on < 0) {
      const char* e = getenv("CVAE_CE_FAST_EVAL");
      fast_on = (e && e[0] == '0') ? 0 : 1;
    }
    const bool fast = !GRAD && !SAVE && fast_on && a.E < 16 * nb;
#define CE_BF3(NB_)                                                                                        \
    do {                                                                                                   \
      if constexpr (!GRAD && !SAVE) {                                                                      \
        if (fast) {                                                                                        \
          hipLaunchKernelGGL((ce_presplit_kernel<NB_>), dim3(pg), dim3(256), 0, s, a.V, a.E, a.W, a.vec, Wp, \
                             1.4426950408889634f, a.bias);                                                 \
          CVAE_LAUNCH_CHECK("ce_presplit_kernel");                                                         \
          hipLaunchKernelGGL((ce_core_bf3_kernel<NB_, false, false, true>), grid, block, 0, s, a, Wp);      \
          break;                                                                                           \
        }                                                                                                  \
      }                                                                                                    \
      hipLaunchKernelGGL((ce_presplit_kernel<NB_>), dim3(pg), dim3(256), 0, s, a.V, a.E, a.W, a.vec, Wp, 1.f, nullptr);  \
      CVAE_LAUNCH_CHECK("ce_presplit_kernel");                                                             \
      hipLaunchKernelGGL((ce_core_bf3_kernel<NB_, GRAD, SAVE>), grid, block, 0, s, a, Wp);                 \
    } while (0)
    if (nb == 1) CE_BF3(1);
    else if (nb == 2) CE_BF3(2);
    else if (nb == 4) CE_BF3(4);
    else if (nb == 7) CE_BF3(7);
    else CE_BF3(13);
#undef CE_BF3
    CVAE_LAUNCH_CHECK("ce_core_bf3_kernel");
    return 0;
  }
  const int nq = (a.E + 7) / 8;
#define CE_CASE(NQ_)                                                                          \
  do {                                                                                        \
    if (a.vec && a.E == 8 * NQ_) hipLaunchKernelGGL((ce_core_kernel<NQ_, GRAD, true, SAVE>), grid, block, 0, s, a);  \
    else hipLaunchKernelGGL((ce_core_kernel<NQ_, GRAD, false, SAVE>), grid, block, 0, s, a);        \
  } while (0)
  if (nq <= 2) CE_CASE(2);
  else if (nq <= 4) CE_CASE(4);
  else if (nq <= 8) CE_CASE(8);
  else if (nq <= 13) CE_CASE(13);
  else CE_CASE(25);
#undef CE_CASE
  CVAE_LAUNCH_CHECK("ce_core_kernel");
  return 0;
}

// Profiling mode only (bench.py's roofline leg): the live-row count, so that the recorded work is
// the ALGORITHMIC one (rows that are multiplied), not the padded N.  Synchronises the stream.
static long live_rows_for_profile(const int* nlive, long N, hipStream_t s) {
  if (!prof_enabled()) return N;
  int h = 0;
  if (hipStreamSynchronize(s) != hipSuccess) return N;
  if (hipMemcpy(&h, nlive, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return N;
  return h;
}

// X [nseg*TR, E]; tgt [TR]; loss, lse [nseg*TR]
static long ce_ldp(int V) { return (long)cdiv(V, CE_TN) * CE_TN; }
// saved logits of the live rows, then the live-row list and its count (the backward pass reuses them: no second scan)
static long ce_logits_only_floats(long N, int V) { return ((N + CE_ROWS - 1) / CE_ROWS * CE_ROWS) * ce_ldp(V); }
long ce_fast_logits_floats(long N, int V) { return ce_logits_only_floats(N, V) + N + 16; }
static int* ce_saved_rows(float* logits, long N, int V) { return reinterpret_cast<int*>(logits + ce_logits_only_floats(N, V)); }

// logits: null, or ce_fast_logits_floats(N, V) floats (16-byte aligned) that receive the logits of the live rows
int launch_ce_fast_forward(const float* X, const float* W, const float* bias, const int64_t* tgt, int64_t pad,
                           long N, long TR, int E, int V, float* loss, float* lse, float* ws, float* logits,
                           hipStream_t s) {
  CVAE_REQUIRE(ce_fast_supported(E) && V > 0 && TR > 0 && N % TR == 0 && N < (1L << 31), "ce_forward: bad shape");
  if (N == 0) return 0;
  CVAE_REQUIRE(X && W && bias && tgt && loss && lse && ws, "ce_forward: null operand");
  const int cap = ce_cap(N, V);
  int* rows = reinterpret_cast<int*>(ws);
  int* counts = rows + N;
  int* nlive = counts + ce_counts_slots(N) - 4;
  float* tl = ws + N + ce_counts_slots(N);
  if (logits) {                                  // the list lives with the saved logits (launch_ce_fast_backward)
    rows = ce_saved_rows(logits, N, V);
    nlive = rows + N;
  }
  CeArgs a = {};
  a.X = X; a.W = W; a.bias = bias; a.tgt = tgt; a.rows = rows; a.nlive = nlive;
  a.N = (int)N; a.TR = (int)TR; a.E = E; a.V = V;
  a.vec = aligned16(X) && aligned16(W) && (E % 4 == 0);
  a.cap = cap; a.pstride = N; a.tl = tl; a.pm = tl + N; a.ps = a.pm + (long)cap * N;
  CVAE_TRY(launch_rows(tgt, pad, N, TR, rows, counts, nlive, loss, lse, tl, s));
  {
    const long live = live_rows_for_profile(nlive, N, s);
    ProfScope prof(PK_CE_FWD, s, 2.0 * live * E * V);
    if (logits) {
      CVAE_REQUIRE(aligned16(logits), "ce_forward: logits buffer must be 16-byte aligned");
      a.P = logits; a.ldp = ce_ldp(V);
      CVAE_TRY((launch_core<false, true>(a, s, ws + ce_fwd_base_floats(N, V))));
    } else {
      CVAE_TRY(launch_core<false>(a, s, ws + ce_fwd_base_floats(N, V)));
    }
  }
  hipLaunchKernelGGL(ce_combine2_kernel, dim3(cdiv(N, 256)), dim3(256), 0, s, a, persistent_grid_blocks(), loss, lse);
  CVAE_LAUNCH_CHECK("ce_combine2_kernel");
  return 0;
}

static long ce_chunk_rows(long N, int V) {
  static long mb = -1;
  if (mb < 0) {
    const char* e = getenv("CVAE_CE_CHUNK_MB");
    mb = e ? atol(e) : 512;
    if (mb < 1) mb = 1;
  }
  long ch = (mb << 20) / (ce_ldp(V) * 4);
  ch = ch / CE_ROWS * CE_ROWS;
  if (ch < CE_ROWS) ch = CE_ROWS;
  const long nr = (N + CE_ROWS - 1) / CE_ROWS * CE_ROWS;
  return ch < nr ? ch : nr;
}
// workspace (floats): rows[N] | counts | nlive | P[chunk * ldp] | split-K slabs
static long ce_bwd_base_floats(long N, int V) {
  return N + ce_counts_slots(N) + 16 + ce_chunk_rows(N, V) * ce_ldp(V) + gemm_wide_part_floats(200, true) + 16;
}
long ce_fast_bwd_workspace_floats(long N, int V) { return ce_bwd_base_floats(N, V) + bf3_presplit_floats(208, V); }

// dX [N,E] overwritten; dW [V,E], db [V] overwritten.
int launch_ce_fast_backward(const float* X, const float* W, const float* bias, const int64_t* tgt, int64_t pad,
                            const float* lse, const float* gsent, long N, long TR, int R, int E, int V,
                            float* dX, float* dW, float* db, float* ws, float* logits, hipStream_t s) {
  CVAE_REQUIRE(ce_fast_supported(E) && V > 0 && TR > 0 && R > 0 && TR % R == 0 && N % TR == 0 && N < (1L << 31),
               "ce_backward: bad shape");
  ZeroBatch zb(s);
  zb.add(dW, (long)V * E);
  zb.add(db, V);
  if (N == 0) return zb.flush();
  CVAE_REQUIRE(X && W && bias && tgt && lse && gsent && dX && dW && db && ws, "ce_backward: null operand");
  zb.add(dX, N * E);
  CVAE_TRY(zb.flush());
  int* rows = reinterpret_cast<int*>(ws);
  int* counts = rows + N;
  int* nlive = counts + ce_counts_slots(N) - 4;
  float* P = ws + N + ce_counts_slots(N) + 16;
  P = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(P) + 15) & ~(uintptr_t)15);
  // logits != null: the forward pass saved the logits of ALL live rows there; they become the gradient in
  // place and the two products run once over all rows (no chunking, nothing recomputed)
  const long ldp = ce_ldp(V), CH = logits ? (N + CE_ROWS - 1) / CE_ROWS * CE_ROWS : ce_chunk_rows(N, V);
  float* part = logits ? P : P + CH * ldp;
  if (logits) {
    CVAE_REQUIRE(aligned16(logits), "ce_backward: logits buffer must be 16-byte aligned");
    P = logits;
  }
  part = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(part) + 15) & ~(uintptr_t)15);
  if (logits) {                                  // the forward pass left the list behind the logits it saved
    rows = ce_saved_rows(logits, N, V);
    nlive = rows + N;
  } else {
    CVAE_TRY(launch_rows(tgt, pad, N, TR, rows, counts, nlive, nullptr, nullptr, nullptr, s));
  }
  CeArgs a = {};
  a.X = X; a.W = W; a.bias = bias; a.tgt = tgt; a.rows = rows; a.nlive = nlive;
  a.N = (int)N; a.TR = (int)TR; a.E = E; a.V = V;
  a.vec = aligned16(X) && aligned16(W) && (E % 4 == 0);
  a.cap = 64; a.lse = lse; a.gsent = gsent; a.R = R; a.P = P; a.ldp = ldp;
  const long live = live_rows_for_profile(nlive, N, s);
  for (long r0 = 0; r0 < N; r0 += CH) {
    const int rc = (int)((N - r0) < CH ? (N - r0) : CH);
    long lrc = live - r0;                    // live rows of this chunk (profiling: exact; else the bound)
    if (lrc > rc) lrc = rc;
    if (lrc < 0) lrc = 0;
    a.r0 = (int)r0; a.rcount = rc;
    if (logits) {
      a.P = P;
      const int nblk = (int)(rc < 8192 ? rc : 8192);
      ProfScope prof(PK_CE_GRAD, s, 8.0 * lrc * V);        // algorithmic bytes: the live rows' logits in, their gradient out
      hipLaunchKernelGGL(ce_grad_from_logits_kernel, dim3(nblk), dim3(256), 0, s, a);
      CVAE_LAUNCH_CHECK("ce_grad_from_logits_kernel");
    } else {
      ProfScope prof(PK_CE_FWD, s, 2.0 * lrc * E * V);
      CVAE_TRY(launch_core<true>(a, s, ws + ce_bwd_base_floats(N, V)));
    }
    G224Args gx = {};      // dX[rows[r0+m], :] += P[m, :] . W
    gx.A = P; gx.lda = ldp; gx.B = W; gx.ldb = E; gx.C = dX; gx.ldc = E;
    gx.M = rc; gx.N = E; gx.K = V; gx.nlive = nlive; gx.r0 = (int)r0; gx.dev_bound = 1; gx.rows = rows; gx.part = part; gx.prof_units = lrc;
    CVAE_TRY(launch_gemm_wide(true, gx, s));
    G224Args gw = {};      // [dW | db] += P^T . [X[rows[r0+k], :] | 1]
    gw.A = P; gw.lda = ldp; gw.B = X; gw.ldb = E; gw.C = dW; gw.ldc = E;
    gw.M = V; gw.N = E; gw.K = rc; gw.nlive = nlive; gw.r0 = (int)r0; gw.dev_bound = 2; gw.rows = rows;
    gw.extra = db; gw.beta = 1; gw.part = part; gw.prof_units = lrc;
    CVAE_TRY(launch_gemm_wide(false, gw, s));
  }
  return 0;
}

// ------------------------------------------------------------------------------------------
// nd projections + CE over the SAME targets (the decoder and the coupled decoder of a training step): one live-row
// list, kept behind the FIRST decoder's saved logits; the backward pass takes it from there, zero-fills all
// gradients in one launch and runs the decoders' products of one kind in one launch where they are small enough
// to share the grid.  Saved-logits path only (logits[d] = ce_fast_logits_floats(N, V) floats each).
int launch_ce_fast_forward_n(int nd, const float* const* X, const float* const* W, const float* const* bias,
                             const int64_t* tgt, int64_t pad, long N, long TR, int E, int V, float* const* loss,
                             float* const* lse, float* const* ws, float* const* logits, hipStream_t s) {
  CVAE_REQUIRE(nd >= 1 && nd <= CeRowsOut::MAX, "ce_forward_n: %d decoders (max %d)", nd, CeRowsOut::MAX);
  CVAE_REQUIRE(ce_fast_supported(E) && V > 0 && TR > 0 && N % TR == 0 && N < (1L << 31), "ce_forward_n: bad shape");
  if (N == 0) return 0;
  CVAE_REQUIRE(X && W && bias && tgt && loss && lse && ws && logits, "ce_forward_n: null operand");
  const int cap = ce_cap(N, V);
  int* rows = ce_saved_rows(logits[0], N, V);
  int* nlive = rows + N;
  CeArgs a[CeRowsOut::MAX] = {};
  CeRowsOut o = {};
  o.n = nd;
  for (int d = 0; d < nd; ++d) {
    CVAE_REQUIRE(X[d] && W[d] && bias[d] && loss[d] && lse[d] && ws[d] && logits[d] && aligned16(logits[d]),
                 "ce_forward_n: null or misaligned operand");
    float* tl = ws[d] + N + ce_counts_slots(N);
    a[d].X = X[d]; a[d].W = W[d]; a[d].bias = bias[d]; a[d].tgt = tgt; a[d].rows = rows; a[d].nlive = nlive;
    a[d].N = (int)N; a[d].TR = (int)TR; a[d].E = E; a[d].V = V;
    a[d].vec = aligned16(X[d]) && aligned16(W[d]) && (E % 4 == 0);
    a[d].cap = cap; a[d].pstride = N; a[d].tl = tl; a[d].pm = tl + N; a[d].ps = a[d].pm + (long)cap * N;
    a[d].P = logits[d]; a[d].ldp = ce_ldp(V);
    o.loss[d] = loss[d]; o.lse[d] = lse[d]; o.tl[d] = tl;
  }
  int* counts = reinterpret_cast<int*>(ws[0]) + N;
  CVAE_TRY(launch_rows_n(tgt, pad, N, TR, rows, counts, nlive, o, s));
  const long live = live_rows_for_profile(nlive, N, s);
  for (int d = 0; d < nd; ++d) {
    {
      ProfScope prof(PK_CE_FWD, s, 2.0 * live * E * V);
      CVAE_TRY((launch_core<false, true>(a[d], s, ws[d] + ce_fwd_base_floats(N, V))));
    }
    hipLaunchKernelGGL(ce_combine2_kernel, dim3(cdiv(N, 256)), dim3(256), 0, s, a[d], persistent_grid_blocks(), loss[d], lse[d]);
    CVAE_LAUNCH_CHECK("ce_combine2_kernel");
  }
  return 0;
}

int launch_ce_fast_backward_n(int nd, const float* const* X, const float* const* W, const float* const* bias,
                              const int64_t* tgt, int64_t pad, const float* const* lse, const float* const* gsent,
                              long N, long TR, int R, int E, int V, float* const* dX, float* const* dW,
                              float* const* db, float* ws, float* const* logits, hipStream_t s) {
  CVAE_REQUIRE(nd >= 1 && nd <= CeRowsOut::MAX && nd <= G224_MAX_BATCH, "ce_backward_n: %d decoders", nd);
  CVAE_REQUIRE(ce_fast_supported(E) && V > 0 && TR > 0 && R > 0 && TR % R == 0 && N % TR == 0 && N < (1L << 31),
               "ce_backward_n: bad shape");
  CVAE_REQUIRE(X && W && bias && tgt && lse && gsent && dX && dW && db && ws && logits, "ce_backward_n: null operand");
  ZeroBatch zb(s);
  for (int d = 0; d < nd; ++d) {
    CVAE_REQUIRE(X[d] && W[d] && bias[d] && lse[d] && gsent[d] && dX[d] && dW[d] && db[d] && logits[d] &&
                 aligned16(logits[d]), "ce_backward_n: null or misaligned operand");
    zb.add(dW[d], (long)V * E);
    zb.add(db[d], V);
    if (N) zb.add(dX[d], N * E);
  }
  CVAE_TRY(zb.flush());
  if (N == 0) return 0;
  int* rows = ce_saved_rows(logits[0], N, V);               // left there by launch_ce_fast_forward_n
  int* nlive = rows + N;
  const long ldp = ce_ldp(V);
  const long live = live_rows_for_profile(nlive, N, s);
  G224Args gx[G224_MAX_BATCH] = {}, gw[G224_MAX_BATCH] = {};
  CeArgs ga[G224_MAX_BATCH] = {};
  for (int d = 0; d < nd; ++d) {
    CeArgs& a = ga[d];
    a.X = X[d]; a.W = W[d]; a.bias = bias[d]; a.tgt = tgt; a.rows = rows; a.nlive = nlive;
    a.N = (int)N; a.TR = (int)TR; a.E = E; a.V = V;
    a.vec = aligned16(X[d]) && aligned16(W[d]) && (E % 4 == 0);
    a.cap = 64; a.lse = lse[d]; a.gsent = gsent[d]; a.R = R; a.P = logits[d]; a.ldp = ldp;
    a.r0 = 0; a.rcount = (int)N;
    // dX[rows[m], :] += P[m, :] . W
    gx[d].A = logits[d]; gx[d].lda = ldp; gx[d].B = W[d]; gx[d].ldb = E; gx[d].C = dX[d]; gx[d].ldc = E;
    gx[d].M = (int)N; gx[d].N = E; gx[d].K = V; gx[d].nlive = nlive; gx[d].r0 = 0; gx[d].dev_bound = 1; gx[d].rows = rows;
    gx[d].prof_units = live;
    // [dW | db] += P^T . [X[rows[k], :] | 1]
    gw[d].A = logits[d]; gw[d].lda = ldp; gw[d].B = X[d]; gw[d].ldb = E; gw[d].C = dW[d]; gw[d].ldc = E;
    gw[d].M = V; gw[d].N = E; gw[d].K = (int)N; gw[d].nlive = nlive; gw[d].r0 = 0; gw[d].dev_bound = 2; gw[d].rows = rows;
    gw[d].extra = db[d]; gw[d].beta = 1; gw[d].prof_units = live;
  }
  auto to_gradient = [&](int d) -> int {
    const int nblk = (int)(N < 8192 ? N : 8192);
    ProfScope prof(PK_CE_GRAD, s, 8.0 * live * V);         // algorithmic bytes: the live rows' logits in, their gradient out
    hipLaunchKernelGGL(ce_grad_from_logits_kernel, dim3(nblk), dim3(256), 0, s, ga[d]);
    CVAE_LAUNCH_CHECK("ce_grad_from_logits_kernel");
    return 0;
  };
  // Small batches: the decoders' products of one kind share a launch (a launch there is a chain of round trips).  Taken
  // while every product still gets two k-ranges per tile on its share of the grid (157 row tiles of a 20 000-word
  // vocabulary on half of the grid would be one k-range each) and an item's chain of chunks stays short (V = 20 000 as
  // the contraction length, dX at the Yelp shape: 357 us batched against 2 x 110 us one after the other).
  const long share = persistent_grid_blocks() / nd;
  auto small = [&](const G224Args& g, long tiles) {
    const long ksb = share / tiles, kunits = cdiv(g.K, 16);
    return nd > 1 && ksb >= 2 && kunits / ksb <= 64;
  };
  if (small(gx[0], cdiv(N, 128)) && small(gw[0], cdiv(V, 128))) {
    for (int d = 0; d < nd; ++d) CVAE_TRY(to_gradient(d));
    CVAE_TRY(launch_gemm_wide_n(true, gx, nd, s));
    CVAE_TRY(launch_gemm_wide_n(false, gw, nd, s));
    return 0;
  }
  // Big batches: decoder by decoder, so that a decoder's gradient w.r.t. its logits (232 MB at the Yelp shape) is
  // still in the Infinity Cache when its two products read it.
  float* part = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(ws) + 15) & ~(uintptr_t)15);
  for (int d = 0; d < nd; ++d) {
    CVAE_TRY(to_gradient(d));
    gx[d].part = part; gw[d].part = part;
    CVAE_TRY(launch_gemm_wide(true, gx[d], s));
    CVAE_TRY(launch_gemm_wide(false, gw[d], s));
  }
  return 0;
}

}  // namespace cvae
