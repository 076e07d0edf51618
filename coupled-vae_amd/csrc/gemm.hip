// fp32 GEMM on the CDNA4 matrix cores (exact-f32 MFMA; gfx950 has no tf32/xf32 path and the
// parity bar is 1e-4 relative, so bf16 operands are not an option for these contractions).
//
//   gemm_tiled   : LDS-staged, BMxBN block tile, v_mfma_f32_32x32x2_f32, 4 waves (2x2).
//                  Used for every "many rows" contraction: input projections over T*B rows,
//                  W_c, weight gradients, the logits chunk of the CE backward.
//   gemm_skinny  : M <= 64 rows per block, operands straight from L2 to registers,
//                  v_mfma_f32_16x16x4_f32, K split over the block's waves and reduced
//                  through LDS.  Used for the per-timestep recurrent products at training
//                  batch sizes (B=32), where one 32x32 tile per wave over the whole K would
//                  serialise 256+ dependent MFMAs; also hosts the fused GRU/LSTM cell
//                  epilogue (skinny.hip).
#include "common.hpp"
#include "tile.hpp"

namespace cvae {

struct TiledArgs {
  const float* A; const float* B; float* C; const float* bias;
  long lda, ldb, ldc;
  int M, N, K;
  int act, beta;
  int vecA, vecB;   // 16-byte vector loads allowed (base aligned, ld % 4 == 0)
  int kchunk;       // K range per blockIdx.z (multiple of BK); gridDim.z > 1 -> atomic accumulate
  const float* row_lse; const float* row_coef; const int64_t* row_tgt;   // ACT_CEGRAD
};

// zk / nz: index and count of the K ranges (split-K) this launch is divided into
template <int BM, int BN, bool TA, bool TB>
__device__ __forceinline__ void gemm_tiled_body(const TiledArgs& p, const int zk, const int nz) {
  // op(A) rows = m: stored [m][k] when !TA (k contiguous) ; op(B) "rows" = n: stored [n][k] when TB
  using LA = TileLoader<BM, !TA>;
  using LB = TileLoader<BN, TB>;
  constexpr int MI = BM / 64, NI = BN / 64;
  __shared__ __attribute__((aligned(16))) float As[BK * LA::LD];
  __shared__ __attribute__((aligned(16))) float Bs[BK * LB::LD];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int wm0 = (wave >> 1) * (BM / 2), wn0 = (wave & 1) * (BN / 2);

  f32x16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  LA la; LB lb;
  const int kbeg = zk * p.kchunk;
  const int kend = min(p.K, kbeg + p.kchunk);
  const int nk = (kend - kbeg + BK - 1) / BK;
  la.load(p.A, p.lda, m0, p.M, kbeg, kend, p.vecA, tid);
  lb.load(p.B, p.ldb, n0, p.N, kbeg, kend, p.vecB, tid);
  la.store(As, tid);
  lb.store(Bs, tid);
  __syncthreads();

  const int l31 = lane & 31, lhi = lane >> 5;
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) {
      la.load(p.A, p.lda, m0, p.M, kbeg + (kt + 1) * BK, kend, p.vecA, tid);
      lb.load(p.B, p.ldb, n0, p.N, kbeg + (kt + 1) * BK, kend, p.vecB, tid);
    }
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      float a[MI], b[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i) a[i] = As[(kk + lhi) * LA::LD + wm0 + i * 32 + l31];
#pragma unroll
      for (int j = 0; j < NI; ++j) b[j] = Bs[(kk + lhi) * LB::LD + wn0 + j * 32 + l31];
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
    if (kt + 1 < nk) {
      la.store(As, tid);
      lb.store(Bs, tid);
      __syncthreads();
    }
  }

  // epilogue: C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int n = n0 + wn0 + j * 32 + l31;
      if (n >= p.N) continue;
      const float bv = (p.bias && zk == 0) ? p.bias[n] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        if (m < p.M) {
          float v = acc[i][j][r] + bv;
          float* c = p.C + (long)m * p.ldc + n;
          if (nz > 1) {
            atomicAdd(c, v);          // split-K: C was zeroed (or holds the beta term) by the host
          } else {
            if (p.act == ACT_CEGRAD)  // d(loss)/d(logit) = coef * (softmax - onehot)
              v = p.row_coef[m] * (expf(v - p.row_lse[m]) - ((int64_t)n == p.row_tgt[m] ? 1.f : 0.f));
            else
              v = apply_act(v, p.act);
            if (p.beta) v += *c;
            *c = v;
          }
        }
      }
    }
}

template <int BM, int BN, bool TA, bool TB>
__global__ __launch_bounds__(256) void gemm_tiled_kernel(TiledArgs p) {
  gemm_tiled_body<BM, BN, TA, TB>(p, blockIdx.z, gridDim.z);
}

// C = act(A W^T + bias) with BOTH operands k-contiguous (A[m][k], W[n][k]) and K a multiple of 16: the
// main loop of rnn_large_step_kernel (LDS rows of 16 + 4 floats, conflict-free b128 fragment reads, lane
// (l31, lhi) takes k = 8q + 4*lhi + (0..3) of its row on both sides, double-buffered, one barrier per
// 16-wide chunk, 40 KB of LDS: three blocks per CU) with a plain epilogue.  Block tile 128 x 128: wave w
// owns rows 32w .. 32w+31 and four 32-column blocks.  Rows / columns past the edge load a clamped row and
// are not stored.  The x-side projections of the decoder (K = 2E = 400) and W_c (K = Hd) take this path.
constexpr int NT_BK = 16, NT_LD = NT_BK + 4;
__device__ __forceinline__ void gemm_nt_body(const TiledArgs& p) {
  __shared__ __attribute__((aligned(16))) float As[2][128 * NT_LD];
  __shared__ __attribute__((aligned(16))) float Bs[2][128 * NT_LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int m0 = blockIdx.y * 128, n0 = blockIdx.x * 128;
  const float* __restrict__ gA = p.A;
  const float* __restrict__ gB = p.B;
  long oa[2], ob[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int idx = tid + 256 * i;                       // piece: row idx / 4, float4 idx % 4 of the chunk
    oa[i] = (long)min(m0 + (idx >> 2), p.M - 1) * p.lda + 4 * (idx & 3);
    ob[i] = (long)min(n0 + (idx >> 2), p.N - 1) * p.ldb + 4 * (idx & 3);
  }
  int kload = 0;
  f32x4 sa[2], sb[2];
#define NT_STAGE_LOAD()                                                                \
  do {                                                                                 \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                    \
      sa[i] = *reinterpret_cast<const f32x4*>(gA + oa[i] + kload);                     \
      sb[i] = *reinterpret_cast<const f32x4*>(gB + ob[i] + kload);                     \
    }                                                                                  \
    kload += NT_BK;                                                                    \
  } while (0)
#define NT_STAGE_STORE(buf)                                                            \
  do {                                                                                 \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                    \
      const int idx = tid + 256 * i;                                                   \
      *reinterpret_cast<f32x4*>(&As[buf][(idx >> 2) * NT_LD + 4 * (idx & 3)]) = sa[i]; \
      *reinterpret_cast<f32x4*>(&Bs[buf][(idx >> 2) * NT_LD + 4 * (idx & 3)]) = sb[i]; \
    }                                                                                  \
  } while (0)
  f32x16 acc[4];
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;
  const int nk = p.K / NT_BK;
  NT_STAGE_LOAD();
  NT_STAGE_STORE(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) NT_STAGE_LOAD();
    const float* ar = &As[buf][(wave * 32 + l31) * NT_LD + 4 * lhi];
    const float* br = &Bs[buf][l31 * NT_LD + 4 * lhi];
#pragma unroll
    for (int q = 0; q < NT_BK / 8; ++q) {
      const float4 a = *reinterpret_cast<const float4*>(ar + 8 * q);
      float4 b[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) b[g] = *reinterpret_cast<const float4*>(br + g * 32 * NT_LD + 8 * q);
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b[g].x, acc[g], 0, 0, 0);
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b[g].y, acc[g], 0, 0, 0);
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b[g].z, acc[g], 0, 0, 0);
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b[g].w, acc[g], 0, 0, 0);
    }
    if (kt + 1 < nk) NT_STAGE_STORE(buf ^ 1);
    __syncthreads();
  }
#undef NT_STAGE_LOAD
#undef NT_STAGE_STORE
  // acc[g][r] = C[m0 + 32*wave + (r&3) + 8*(r>>2) + 4*lhi][n0 + 32*g + l31]
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int n = n0 + 32 * g + l31;
    if (n >= p.N) continue;
    const float bv = p.bias ? p.bias[n] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * lhi;
      if (m < p.M) {
        float* c = p.C + (long)m * p.ldc + n;
        float v = apply_act(acc[g][r] + bv, p.act);
        if (p.beta) v += *c;
        *c = v;
      }
    }
  }
}
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(TiledArgs p) { gemm_nt_body(p); }

// The same product on the bf16 matrix pipe, error-compensated ("bf16x3": both operands split into bf16 hi + lo while
// they are staged, hi.lo + lo.hi + hi.hi per 16-deep K block, fp32 accumulation; the forward products' form, see
// ce_fast.hip / rnn_large.hip, whose main loop this is).  K chunk 32, LDS rows of four 16-byte slots XOR-swizzled with
// (row >> 2) & 3, two register sets (chunk kt+2 is fetched while chunk kt is multiplied), 64 KB of LDS.  K a multiple
// of 8: the slots of a last partial chunk are zero-filled.
typedef __bf16 nt_bf16x8 __attribute__((ext_vector_type(8)));
typedef float nt_f32x8 __attribute__((ext_vector_type(8)));
constexpr int NTB_BK = 32;
// Tile of this workgroup.  With two column tiles (the decoders' W_c product: N = 200) the pair that shares 128 rows of A
// is placed 8 workgroups apart -- consecutive workgroups go round the 8 XCDs, each with its own L2 -- so the second
// read of those rows (3 GB per evaluation batch) is an L2 hit instead of a second trip to HBM.
__device__ __forceinline__ void nt_tile_xy(int& bx, int& by) {
  bx = blockIdx.x; by = blockIdx.y;
  if (gridDim.x == 2) {
    const int P = gridDim.y, L = blockIdx.x + 2 * blockIdx.y, full = (P / 8) * 16;
    if (L < full) {
      const int r = L & 15;
      bx = r >> 3; by = (L >> 4) * 8 + (r & 7);
    } else {
      const int rem = P - (P / 8) * 8, r = L - full;
      bx = r / rem; by = (P / 8) * 8 + r % rem;
    }
  }
}
__device__ __forceinline__ void gemm_nt_bf3_body(const TiledArgs& p) {
  constexpr int A_SL = 128 * 4, B_SL = 128 * 4, BUF_SL = 2 * A_SL + 2 * B_SL;   // slots: [A hi][A lo][B hi][B lo]
  extern __shared__ __attribute__((aligned(16))) float nt_smem[];
  nt_bf16x8* sm = reinterpret_cast<nt_bf16x8*>(nt_smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  int bx, by;
  nt_tile_xy(bx, by);
  const int m0 = by * 128, n0 = bx * 128;
  const float* __restrict__ gA = p.A;
  const float* __restrict__ gB = p.B;
  long oa[2], ob[2];
  int slot[2], ksl[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int idx = tid + 256 * i, r = idx >> 2, sl = idx & 3;       // task: row r, 8 k from 8*sl
    oa[i] = (long)min(m0 + r, p.M - 1) * p.lda + 8 * sl;
    ob[i] = (long)min(n0 + r, p.N - 1) * p.ldb + 8 * sl;
    slot[i] = r * 4 + (sl ^ ((r >> 2) & 3));
    ksl[i] = 8 * sl;
  }
  int kload = 0;
  f32x4 ra[2][2][2], rb[2][2][2];
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#define NTB_LOAD(set)                                                                            \
  do {                                                                                           \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                              \
      const bool in = kload + ksl[i] < p.K;                                                      \
      ra[set][i][0] = in ? *reinterpret_cast<const f32x4*>(gA + oa[i] + kload) : z4;             \
      ra[set][i][1] = in ? *reinterpret_cast<const f32x4*>(gA + oa[i] + kload + 4) : z4;         \
      rb[set][i][0] = in ? *reinterpret_cast<const f32x4*>(gB + ob[i] + kload) : z4;             \
      rb[set][i][1] = in ? *reinterpret_cast<const f32x4*>(gB + ob[i] + kload + 4) : z4;         \
    }                                                                                            \
    kload += NTB_BK;                                                                             \
  } while (0)
#define NTB_SPLIT(x0, x1, hi, lo)                                                                \
  do {                                                                                           \
    const nt_f32x8 v_ = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};                \
    hi = __builtin_convertvector(v_, nt_bf16x8);                                                 \
    lo = __builtin_convertvector(v_ - __builtin_convertvector(hi, nt_f32x8), nt_bf16x8);         \
  } while (0)
#define NTB_STORE(set, buf)                                                                      \
  do {                                                                                           \
    nt_bf16x8* base = sm + (buf) * BUF_SL;                                                       \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                              \
      nt_bf16x8 hi, lo;                                                                          \
      NTB_SPLIT(ra[set][i][0], ra[set][i][1], hi, lo);                                           \
      base[slot[i]] = hi;                                                                        \
      base[A_SL + slot[i]] = lo;                                                                 \
      NTB_SPLIT(rb[set][i][0], rb[set][i][1], hi, lo);                                           \
      base[2 * A_SL + slot[i]] = hi;                                                             \
      base[2 * A_SL + B_SL + slot[i]] = lo;                                                      \
    }                                                                                            \
  } while (0)
  f32x16 acc[4];
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;
  const int arow = wave * 32 + l31, sw = (l31 >> 2) & 3;
  const int nk = (p.K + NTB_BK - 1) / NTB_BK;
  NTB_LOAD(0);
  NTB_STORE(0, 0);
  if (nk > 1) NTB_LOAD(1);
  __syncthreads();
  for (int kt0 = 0; kt0 < nk; kt0 += 2) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int kt = kt0 + u;
      if (kt < nk) {
        if (kt + 2 < nk) NTB_LOAD(u);
        const nt_bf16x8* base = sm + u * BUF_SL;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
          const int sl = (2 * kb + lhi) ^ sw;
          const nt_bf16x8 ah = base[arow * 4 + sl], al = base[A_SL + arow * 4 + sl];
          nt_bf16x8 bh[4], bl[4];
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            bh[g] = base[2 * A_SL + (g * 32 + l31) * 4 + sl];
            bl[g] = base[2 * A_SL + B_SL + (g * 32 + l31) * 4 + sl];
          }
#pragma unroll
          for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[g], acc[g], 0, 0, 0);
#pragma unroll
          for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[g], acc[g], 0, 0, 0);
#pragma unroll
          for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[g], acc[g], 0, 0, 0);
        }
        if (kt + 1 < nk) NTB_STORE(u ^ 1, u ^ 1);
        __syncthreads();
      }
    }
  }
#undef NTB_LOAD
#undef NTB_SPLIT
#undef NTB_STORE
  // acc[g][r] = C[m0 + 32*wave + (r&3) + 8*(r>>2) + 4*lhi][n0 + 32*g + l31]
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int n = n0 + 32 * g + l31;
    if (n >= p.N) continue;
    const float bv = p.bias ? p.bias[n] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * lhi;
      if (m < p.M) {
        float* c = p.C + (long)m * p.ldc + n;
        float v = apply_act(acc[g][r] + bv, p.act);
        if (p.beta) v += *c;
        *c = v;
      }
    }
  }
}
__global__ __launch_bounds__(256, 2) void gemm_nt_bf3_kernel(TiledArgs p) { gemm_nt_bf3_body(p); }
constexpr size_t NTB_LDS = (size_t)2 * (4 * 128 * 4) * 16;      // 64 KB
static bool gemm_nt_enabled() {
  static int on = -1;
  if (on < 0) { const char* e = getenv("CVAE_GEMM_NT"); on = (e && e[0] == '0') ? 0 : 1; }
  return on != 0;
}
// plain A, transposed B, K % 16 == 0, 16-byte aligned rows, no CE-gradient epilogue
static bool gemm_nt_ok(const GemmDesc& d) {
  return gemm_nt_enabled() && !d.transA && d.transB && d.K >= NT_BK && d.K % NT_BK == 0 && d.act != ACT_CEGRAD &&
         aligned16(d.A) && aligned16(d.B) && d.lda % 4 == 0 && d.ldb % 4 == 0;
}
// the split-bf16 form: on by default (cvae_split_bf16), K a multiple of 8
static bool gemm_nt_bf3_ok(const GemmDesc& d) {
  return gemm_nt_enabled() && split_bf16_enabled() && !d.transA && d.transB && d.K >= 32 && d.K % 8 == 0 &&
         d.act != ACT_CEGRAD && aligned16(d.A) && aligned16(d.B) && d.lda % 4 == 0 && d.ldb % 4 == 0;
}

// Batch of products with one B: blockIdx.z = batch; A and C advance by a stride, the row count of
// every batch comes with the launch (the IW-eval decoder: one slab of live rows per timestep).
struct TiledBatch { long sA, sC; int m[GEMM_MAX_BATCH]; };
template <int BM, int BN, bool TA, bool TB>
__global__ __launch_bounds__(256) void gemm_tiled_batched_kernel(TiledArgs p, TiledBatch b) {
  p.M = b.m[blockIdx.z];
  if ((int)blockIdx.y * BM >= p.M) return;
  p.A += (long)blockIdx.z * b.sA;
  p.C += (long)blockIdx.z * b.sC;
  gemm_tiled_body<BM, BN, TA, TB>(p, 0, 1);
}

__global__ __launch_bounds__(256, 2) void gemm_nt_batched_kernel(TiledArgs p, TiledBatch b) {
  p.M = b.m[blockIdx.z];
  if ((int)blockIdx.y * 128 >= p.M) return;
  p.A += (long)blockIdx.z * b.sA;
  p.C += (long)blockIdx.z * b.sC;
  gemm_nt_body(p);
}
__global__ __launch_bounds__(256, 2) void gemm_nt_bf3_batched_kernel(TiledArgs p, TiledBatch b) {
  p.M = b.m[blockIdx.z];
  int bx, by;
  nt_tile_xy(bx, by);
  if (by * 128 >= p.M) return;
  p.A += (long)blockIdx.z * b.sA;
  p.C += (long)blockIdx.z * b.sC;
  gemm_nt_bf3_body(p);
}
static int gemm_nt_bf3_attr() {
  static bool done = false;
  if (!done) {
    CVAE_TRY(check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_bf3_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024), "gemm_nt_bf3 LDS attribute"));
    CVAE_TRY(check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_bf3_batched_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024), "gemm_nt_bf3 LDS attribute"));
    done = true;
  }
  return 0;
}

static int launch_tiled_batched(const GemmDesc& d, hipStream_t s) {
  CVAE_REQUIRE(d.nbatch >= 1 && d.nbatch <= GEMM_MAX_BATCH && d.host_mbatch, "gemm(batched): 1..%d batches", GEMM_MAX_BATCH);
  CVAE_REQUIRE(!d.transA && d.act != ACT_CEGRAD && !d.colsum_out && !d.rows, "gemm(batched): unsupported form");
  TiledArgs a = {};
  a.A = d.A; a.B = d.B; a.C = d.C; a.bias = d.bias;
  a.lda = d.lda; a.ldb = d.ldb; a.ldc = d.ldc;
  a.N = d.N; a.K = d.K; a.act = d.act; a.beta = d.beta;
  a.vecA = aligned16(d.A) && (d.lda % 4 == 0) && (d.batchA % 4 == 0);
  a.vecB = aligned16(d.B) && (d.ldb % 4 == 0);
  a.kchunk = (d.K + BK - 1) / BK * BK;
  TiledBatch b = {};
  b.sA = d.batchA; b.sC = d.batchC;
  int mmax = 0;
  double rows = 0;
  for (int i = 0; i < d.nbatch; ++i) {
    CVAE_REQUIRE(d.host_mbatch[i] >= 0, "gemm(batched): negative row count");
    b.m[i] = d.host_mbatch[i];
    mmax = b.m[i] > mmax ? b.m[i] : mmax;
    rows += b.m[i];
  }
  if (mmax == 0) return 0;
  constexpr int BM = 128, BN = 128;
  dim3 grid(cdiv(d.N, BN), cdiv(mmax, BM), d.nbatch);
  const bool nt = (gemm_nt_ok(d) || gemm_nt_bf3_ok(d)) && d.batchA % 4 == 0;
  ProfScope prof(nt ? PK_GEMM_NT : PK_GEMM_TILED, s, 2.0 * rows * d.N * d.K);
  if (nt && gemm_nt_bf3_ok(d)) {
    CVAE_TRY(gemm_nt_bf3_attr());
    hipLaunchKernelGGL(gemm_nt_bf3_batched_kernel, grid, dim3(256), NTB_LDS, s, a, b);
  } else if (nt) hipLaunchKernelGGL(gemm_nt_batched_kernel, grid, dim3(256), 0, s, a, b);
  else if (d.transB) hipLaunchKernelGGL((gemm_tiled_batched_kernel<BM, BN, false, true>), grid, dim3(256), 0, s, a, b);
  else hipLaunchKernelGGL((gemm_tiled_batched_kernel<BM, BN, false, false>), grid, dim3(256), 0, s, a, b);
  CVAE_LAUNCH_CHECK("gemm_tiled_batched");
  return 0;
}

template <int BM, int BN>
static int launch_tiled(const GemmDesc& d, hipStream_t s) {
  TiledArgs a;
  a.A = d.A; a.B = d.B; a.C = d.C; a.bias = d.bias;
  a.lda = d.lda; a.ldb = d.ldb; a.ldc = d.ldc;
  a.M = d.M; a.N = d.N; a.K = d.K; a.act = d.act; a.beta = d.beta;
  a.vecA = aligned16(d.A) && (d.lda % 4 == 0);
  a.vecB = aligned16(d.B) && (d.ldb % 4 == 0);
  a.row_lse = d.row_lse; a.row_coef = d.row_coef; a.row_tgt = d.row_tgt;
  // split-K when the output has too few tiles to fill 256 CUs and K is long
  const long tiles = (long)cdiv(d.N, BN) * cdiv(d.M, BM);
  int ksplit = 1;
  if (d.act == ACT_NONE && tiles < 128 && d.K >= 1024 && !deterministic_enabled()) {     // (atomic accumulation)
    ksplit = (int)((512 + tiles - 1) / tiles);
    const int maxs = d.K / 256;
    if (ksplit > maxs) ksplit = maxs;
    if (ksplit < 1) ksplit = 1;
  }
  a.kchunk = ((d.K + ksplit - 1) / ksplit + BK - 1) / BK * BK;
  ksplit = (d.K + a.kchunk - 1) / a.kchunk;
  if (ksplit < 1) ksplit = 1;
  if (ksplit > 1 && !d.beta) {
    if (d.ldc == d.N) {
      CVAE_TRY(launch_zero(d.C, (long)d.M * d.N, s));
    } else {
      CVAE_TRY(check_hip(hipMemset2DAsync(d.C, d.ldc * sizeof(float), 0, (size_t)d.N * sizeof(float), d.M, s),
                         "gemm split-K memset2D"));
    }
  }
  dim3 grid(cdiv(d.N, BN), cdiv(d.M, BM), ksplit);
  const bool nt = BM == 128 && BN == 128 && ksplit == 1 && (gemm_nt_ok(d) || gemm_nt_bf3_ok(d));
  ProfScope prof(nt ? PK_GEMM_NT : PK_GEMM_TILED, s, 2.0 * d.M * d.N * d.K);
  if (nt) {
    if (gemm_nt_bf3_ok(d)) {
      CVAE_TRY(gemm_nt_bf3_attr());
      hipLaunchKernelGGL(gemm_nt_bf3_kernel, grid, dim3(256), NTB_LDS, s, a);
    } else {
      hipLaunchKernelGGL(gemm_nt_kernel, grid, dim3(256), 0, s, a);
    }
    CVAE_LAUNCH_CHECK("gemm_nt");
    return 0;
  }
  if (!d.transA && d.transB) hipLaunchKernelGGL((gemm_tiled_kernel<BM, BN, false, true>), grid, dim3(256), 0, s, a);
  else if (!d.transA && !d.transB) hipLaunchKernelGGL((gemm_tiled_kernel<BM, BN, false, false>), grid, dim3(256), 0, s, a);
  else if (d.transA && !d.transB) hipLaunchKernelGGL((gemm_tiled_kernel<BM, BN, true, false>), grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL((gemm_tiled_kernel<BM, BN, true, true>), grid, dim3(256), 0, s, a);
  CVAE_LAUNCH_CHECK("gemm_tiled");
  return 0;
}

int launch_gemm(const GemmDesc& d, hipStream_t s) {
  CVAE_REQUIRE(d.M >= 0 && d.N >= 0 && d.K >= 0, "gemm: negative dimension");
  if (d.nbatch > 0) {
    CVAE_REQUIRE(d.A && d.B && d.C && d.K > 0 && d.N > 0, "gemm(batched): null operand or empty product");
    return launch_tiled_batched(d, s);
  }
  if (d.M == 0 || d.N == 0) return 0;
  CVAE_REQUIRE(d.A && d.B && d.C, "gemm: null operand");
  CVAE_REQUIRE(d.act != ACT_CEGRAD || (d.row_lse && d.row_coef && d.row_tgt && !d.beta), "gemm: CE-grad epilogue operands missing");
  if (d.K == 0) {
    CVAE_REQUIRE(d.act == ACT_NONE && !d.bias, "gemm: K == 0 with an epilogue");
    if (!d.beta) CVAE_TRY(check_hip(hipMemset2DAsync(d.C, d.ldc * sizeof(float), 0, (size_t)d.N * sizeof(float), d.M, s), "gemm K=0 memset"));
    return 0;
  }
  CVAE_REQUIRE(d.lda >= (d.transA ? d.M : d.K) && d.ldb >= (d.transB ? d.K : d.N) && d.ldc >= d.N,
               "gemm: leading dimension smaller than the row length");
  if (d.colsum_out) {
    CVAE_REQUIRE(d.transA && !d.beta, "gemm: colsum_out needs a transA product without accumulation");
    const bool wide_ok = !d.transB && d.act == ACT_NONE && !d.bias && d.K >= 512 && d.N <= 1024 && d.M >= 64 &&
                         gemm_wide_has_spare_column(d.N);
    if (!wide_ok) {                       // plain pass over A, then the product without the extra output
      CVAE_TRY(launch_zero(d.colsum_out, d.M, s));
      CVAE_TRY(launch_colsum_acc(d.A, d.K, d.M, d.lda, d.colsum_out, s));
      GemmDesc e = d;
      e.colsum_out = nullptr;
      return launch_gemm(e, s);
    }
  }
  if (d.rows) {
    CVAE_REQUIRE(d.nlive && !d.transB && d.act == ACT_NONE && !d.bias &&
                 ((d.bound_dim == 1 && !d.transA) || (d.bound_dim == 2 && d.transA)),
                 "gemm: live-row list with an unsupported product form");
    G224Args w = {};
    w.A = d.A; w.lda = d.lda; w.B = d.B; w.ldb = d.ldb; w.C = d.C; w.ldc = d.ldc;
    w.M = d.M; w.N = d.N; w.K = d.K; w.beta = d.beta;
    w.rows = d.rows; w.nlive = d.nlive; w.r0 = 0; w.dev_bound = d.bound_dim; w.a_ind = 1;
    if (d.colsum_out) {
      if (gemm_wide_has_spare_column(d.N)) {
        w.extra = d.colsum_out;
      } else {
        CVAE_TRY(launch_zero(d.colsum_out, d.M, s));
        CVAE_TRY(launch_colsum_acc(d.A, d.K, d.M, d.lda, d.colsum_out, s));
      }
    }
    return launch_gemm_wide(!d.transA, w, s);
  }
  // Skinny path: few rows, plain A, K long enough to be worth splitting over waves.
  if (!d.transA && d.M <= 64 && d.K >= 64 && d.act != ACT_CEGRAD) {
    SkinnyArgs a = {};
    a.A = d.A; a.lda = d.lda; a.B = d.B; a.ldb = d.ldb; a.transB = d.transB;
    a.M = d.M; a.N = d.N; a.K = d.K; a.ngroups = 1; a.gstride = 0;
    a.C = d.C; a.ldc = d.ldc; a.bias = d.bias; a.act = d.act; a.beta = d.beta;
    return launch_skinny(a, EPI_LINEAR, s);
  }
  // Long-K products with a k-major B and at most a few hundred columns (weight gradients over
  // T*B rows, input gradients): persistent split-K kernel of gemm_wide.hip.
  // (wider outputs too when the split-bf16 kernel runs them: the 2048-column weight gradients of the scale-up shape)
  const int wide_max_n = split_bf16_enabled() ? 4096 : 1024;
  if (!d.transB && d.act == ACT_NONE && !d.bias && d.K >= 512 && d.N <= wide_max_n && d.M >= 64) {
    G224Args w = {};
    w.A = d.A; w.lda = d.lda; w.B = d.B; w.ldb = d.ldb; w.C = d.C; w.ldc = d.ldc;
    w.M = d.M; w.N = d.N; w.K = d.K; w.beta = d.beta;
    w.extra = d.colsum_out;
    return launch_gemm_wide(!d.transA, w, s);
  }
  const long tiles128 = (long)cdiv(d.M, 128) * cdiv(d.N, 128);
  if (tiles128 >= 192) return launch_tiled<128, 128>(d, s);
  return launch_tiled<64, 64>(d, s);
}

// The gemm_wide form of a product that launch_gemm runs there without any helper launch (else false)
static bool gemm_wide_form(const GemmDesc& d, G224Args* w) {
  if (d.nbatch > 0 || d.M <= 0 || d.N <= 0 || d.K <= 0 || !d.A || !d.B || !d.C) return false;
  if (d.transB || d.act != ACT_NONE || d.bias) return false;
  if (d.lda < (d.transA ? d.M : d.K) || d.ldb < d.N || d.ldc < d.N) return false;
  if (d.colsum_out && !(d.transA && !d.beta && gemm_wide_has_spare_column(d.N))) return false;
  *w = G224Args{};
  w->A = d.A; w->lda = d.lda; w->B = d.B; w->ldb = d.ldb; w->C = d.C; w->ldc = d.ldc;
  w->M = d.M; w->N = d.N; w->K = d.K; w->beta = d.beta; w->extra = d.colsum_out;
  if (d.rows) {
    if (!d.nlive || !((d.bound_dim == 1 && !d.transA) || (d.bound_dim == 2 && d.transA))) return false;
    w->rows = d.rows; w->nlive = d.nlive; w->r0 = 0; w->dev_bound = d.bound_dim; w->a_ind = 1;
    return true;
  }
  if (!d.transA && d.M <= 64 && d.K >= 64) return false;                       // the skinny path's
  const int wide_max_n = split_bf16_enabled() ? 4096 : 1024;
  if (d.colsum_out && d.N > 1024) return false;
  return d.K >= 512 && d.N <= wide_max_n && d.M >= 64;
}

// n products; those of one shape and form that gemm_wide takes run as ONE launch (launch_gemm_wide_n), anything else
// one by one.  The two decoders' and the two encoder directions' weight- and input-gradient products come in such sets.
int launch_gemm_n(const GemmDesc* dd, int n, hipStream_t s) {
  CVAE_REQUIRE(dd && n >= 1, "gemm_n: no products");
  static int on = -1;
  if (on < 0) { const char* e = getenv("CVAE_WIDE_BATCH"); on = (e && e[0] == '0') ? 0 : 1; }
  G224Args w[G224_MAX_BATCH];
  bool same = on && n >= 2 && n <= G224_MAX_BATCH;
  for (int i = 0; same && i < n; ++i) {
    const GemmDesc& d = dd[i];
    same = gemm_wide_form(d, &w[i]) && d.transA == dd[0].transA && d.M == dd[0].M && d.N == dd[0].N && d.K == dd[0].K &&
           d.beta == dd[0].beta && (d.colsum_out != nullptr) == (dd[0].colsum_out != nullptr) &&
           (d.rows != nullptr) == (dd[0].rows != nullptr) && d.bound_dim == dd[0].bound_dim;
    for (int j = 0; same && j < i; ++j) same = d.C != dd[j].C;                  // distinct outputs
  }
  if (same) return launch_gemm_wide_n(!dd[0].transA, w, n, s);
  for (int i = 0; i < n; ++i) CVAE_TRY(launch_gemm(dd[i], s));
  return 0;
}

}  // namespace cvae
