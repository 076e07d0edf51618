// Split-K GEMM for "few hundred columns" outputs:  C[M,N] += A[M,K] . B[K,N],  B stored [k][n].
// These are the long-K contractions of the training step: the two products that consume the
// softmax gradient (dX = P W, [dW | db] += P^T [X | 1]) and the weight gradients of the
// recurrent layers (dW = dG^T X over T*B rows).
//
//  * persistent grid (2 blocks / CU); the (m-tile, n-tile, k-range) work items are laid out on
//    the DEVICE, because for the CE products the live-row count that bounds M or K only exists
//    in device memory (ce_fast.hip);
//  * block tile 128 x (NF*32) columns, 4 waves x 32 rows, v_mfma_f32_32x32x2_f32; BK = 16,
//    double-buffered LDS, one barrier per chunk, fragments for MFMA step s+1 are read while
//    step s issues;
//  * A is read k-contiguous ([m][k], float4 per lane = 4 MFMA steps) or k-major ([k][m]);
//    B fragments are read with one b128 (+ one b64 + one b32 when NF = 7) per step: the column
//    permutation this implies is undone in the epilogue;
//  * no atomics (13M float atomics cost ~180 us on MI355X, more than the product itself): with
//    one k-range per tile the block stores C directly, otherwise every item stores its partial
//    tile to a slab and gemm_wide_reduce_kernel adds the slabs in a fixed order (deterministic).
#include "common.hpp"
#include "misc.hpp"

namespace cvae {

constexpr int G2_BM = 128, G2_BK = 16, G2_LDA = 20;

int persistent_grid_blocks() {
  static int nblk = 0;
  if (!nblk) {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
      hipDeviceProp_t prop;
      if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
    }
    nblk = 2 * cus;
  }
  return nblk;
}

__device__ __forceinline__ float4 ld4g(const float* __restrict__ src, int k, int K, int vec) {
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

// blocks of one gemm_wide launch: CVAE_WIDE_BLOCKS_PER_CU (default 2) per CU
static int wide_grid_blocks() {
  static int nblk = 0;
  if (!nblk) {
    const char* e = getenv("CVAE_WIDE_BLOCKS_PER_CU");
    int per = e ? atoi(e) : 2;
    if (per < 1 || per > 3) per = 2;
    nblk = persistent_grid_blocks() / 2 * per;
  }
  return nblk;
}

// Work layout, computed identically by the GEMM blocks and by the reduction kernel.
struct WideLayout { int Mlim, Klim, mt, ks, kchunk; };
__device__ __forceinline__ WideLayout wide_layout(const G224Args& p, int nblk) {
  WideLayout w;
  w.Mlim = p.M; w.Klim = p.K;
  if (p.dev_bound) {
    int lim = *p.nlive - p.r0;
    if (lim < 0) lim = 0;
    if (p.dev_bound == 1) w.Mlim = min(w.Mlim, lim); else w.Klim = min(w.Klim, lim);
  }
  w.mt = (w.Mlim + G2_BM - 1) / G2_BM;
  const int kunits = (w.Klim + G2_BK - 1) / G2_BK;
  int ks = (w.mt > 0) ? nblk / (w.mt * p.ntiles) : 1;
  if (ks > kunits / p.min_chunks) ks = kunits / p.min_chunks;      // at least min_chunks chunks per item
  if (ks < 1) ks = 1;
  w.kchunk = ((kunits + ks - 1) / ks) * G2_BK;
  w.ks = (w.kchunk > 0) ? (w.Klim + w.kchunk - 1) / w.kchunk : 1;
  return w;
}

// NF = 4: 128-column tiles; NF = 7: 224-column tiles (fragments 4,5 from a b64, 6 from a b32)
template <bool AKC, int NF>
__global__ __launch_bounds__(256, 2) void gemm_wide_kernel(G224Batch batch) {
  const G224Args& p = batch.a[blockIdx.y];       // one product per grid row (launch_gemm_wide_n)
  constexpr int BN = NF * 32;
  __shared__ __attribute__((aligned(16))) float As[2][AKC ? G2_BM * G2_LDA : G2_BK * G2_BM];
  __shared__ __attribute__((aligned(16))) float Bs[2][G2_BK * BN];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const WideLayout L = wide_layout(p, gridDim.x);
  const int Mlim = L.Mlim, Klim = L.Klim;
  if (Mlim <= 0 || Klim <= 0) return;
  const int mt = L.mt, nt = p.ntiles, ks = L.ks, kchunk = L.kchunk;
  const int items = mt * nt * ks;

  const int tw = p.tile_w;                    // columns per n-tile (multiple of 4, <= BN - 4 when extra)
  const int nf4 = tw >> 2;                    // float4 per staged B row
  const int ones_col = p.extra ? tw : -1;

  // columns of the LDS image that staging never writes: zero, except the ones column
  for (int i = tid; i < 2 * G2_BK * BN; i += 256) {
    const int c = i % BN;
    if (c >= tw) (&Bs[0][0])[i] = (c == ones_col) ? 1.f : 0.f;
  }
  int bkr[4], bc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = tid + 256 * i;
    bkr[i] = (idx < G2_BK * nf4) ? idx / nf4 : -1;
    bc[i] = 4 * (idx % nf4);
  }
  __syncthreads();

  for (int item = blockIdx.x; item < items; item += gridDim.x) {
    const int ntile = item % nt, mtile = (item / nt) % mt, ksp = item / (nt * mt);
    const int m0 = mtile * G2_BM, n0 = ntile * tw;
    const int ncols = min(tw, p.N - n0);
    const int kbeg = ksp * kchunk, kend = min(Klim, kbeg + kchunk);

    float4 sa[2], sb[4];
    // Interior chunks (every load a whole in-bounds float4) take a branch-free path on
    // pointers set up once per item; edge tiles and the K tail take the guarded path.
    // Rows of A past Mlim are clamped to a valid row: their accumulators are never stored.
    const bool fast_item = p.vecA && p.vecB && ncols == tw && (AKC || m0 + G2_BM <= Mlim);
    const float* pa[2];
    const float* pb[4];
    int brn[4];                                // dev_bound == 2: B row index of the next interior chunk
    int arn[2];                                // a_ind, k-major A: A row index of the next interior chunk
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int idx = tid + 256 * i;
      arn[i] = 0;
      if (AKC) {
        const int mm = min(m0 + (idx >> 2), Mlim - 1);
        const long arow = p.a_ind ? (long)p.rows[p.r0 + mm] : (long)mm;
        pa[i] = p.A + arow * p.lda + kbeg + 4 * (idx & 3);
      } else if (p.a_ind) {
        pa[i] = p.A + m0 + 4 * (idx & 31);
        arn[i] = p.rows[p.r0 + min(kbeg + (idx >> 5), Klim - 1)];
      } else {
        pa[i] = p.A + (long)(kbeg + (idx >> 5)) * p.lda + m0 + 4 * (idx & 31);
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int kr = max(bkr[i], 0);
      pb[i] = p.B + ((p.dev_bound == 2) ? 0L : (long)(kbeg + kr) * p.ldb) + n0 + bc[i];
      brn[i] = (p.dev_bound == 2) ? p.rows[p.r0 + min(kbeg + kr, Klim - 1)] : 0;
    }
    const long a_step = AKC ? (long)G2_BK : (long)G2_BK * p.lda;
    const long b_step = (long)G2_BK * p.ldb;
    int knext = kbeg + G2_BK;                  // first k of the chunk after the one being loaded
    // loads the next interior chunk of this item (chunks are taken in order)
    auto fast_load = [&]() {
      if (!AKC && p.a_ind) {
#pragma unroll
        for (int i = 0; i < 2; ++i) sa[i] = *reinterpret_cast<const float4*>(pa[i] + (long)arn[i] * p.lda);
#pragma unroll
        for (int i = 0; i < 2; ++i) arn[i] = p.rows[p.r0 + min(knext + ((tid + 256 * i) >> 5), Klim - 1)];
      } else {
#pragma unroll
        for (int i = 0; i < 2; ++i) { sa[i] = *reinterpret_cast<const float4*>(pa[i]); pa[i] += a_step; }
      }
      if (p.dev_bound == 2) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (bkr[i] >= 0) sb[i] = *reinterpret_cast<const float4*>(pb[i] + (long)brn[i] * p.ldb);
#pragma unroll
        for (int i = 0; i < 4; ++i) brn[i] = p.rows[p.r0 + min(knext + max(bkr[i], 0), Klim - 1)];
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (bkr[i] >= 0) { sb[i] = *reinterpret_cast<const float4*>(pb[i]); pb[i] += b_step; }
      }
      knext += G2_BK;
    };
    auto edge_load = [&](int k0) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int idx = tid + 256 * i;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (AKC) {
          const int row = m0 + (idx >> 2), k = k0 + 4 * (idx & 3);
          if (row < Mlim && k < kend) {
            const long arow = p.a_ind ? (long)p.rows[p.r0 + row] : (long)row;
            v = ld4g(p.A + arow * p.lda + k, k, kend, p.vecA);
          }
        } else {
          const int k = k0 + (idx >> 5), m = m0 + 4 * (idx & 31);
          if (k < kend && m < Mlim) {
            const long arow = p.a_ind ? (long)p.rows[p.r0 + k] : (long)k;
            v = ld4g(p.A + arow * p.lda + m, m, Mlim, p.vecA);
          }
        }
        sa[i] = v;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        const int k = k0 + bkr[i];
        if (bkr[i] >= 0 && k < kend && bc[i] < ncols) {
          const long brow = (p.dev_bound == 2) ? (long)p.rows[p.r0 + k] : (long)k;
          v = ld4g(p.B + brow * p.ldb + n0 + bc[i], bc[i], ncols, p.vecB);
        }
        sb[i] = v;
      }
    };
    auto stage_store = [&](int buf) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int idx = tid + 256 * i;
        if (AKC) *reinterpret_cast<float4*>(&As[buf][(idx >> 2) * G2_LDA + 4 * (idx & 3)]) = sa[i];
        else *reinterpret_cast<float4*>(&As[buf][(idx >> 5) * G2_BM + 4 * (idx & 31)]) = sa[i];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (bkr[i] >= 0) *reinterpret_cast<float4*>(&Bs[buf][bkr[i] * BN + bc[i]]) = sb[i];
    };

    f32x16 acc[NF];
#pragma unroll
    for (int j = 0; j < NF; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    const int nk = (kend - kbeg + G2_BK - 1) / G2_BK;
    const int nfast = fast_item ? (kend - kbeg) / G2_BK : 0;   // interior chunks form a prefix
    auto compute = [&](int buf) {
      // the chunk's 8 MFMA steps: (q, s) uses k = 8q + 4*lhi + s
      float a[8];
      if (AKC) {
        const float* ar = &As[buf][(wave * 32 + l31) * G2_LDA + 4 * lhi];
        const float4 a0 = *reinterpret_cast<const float4*>(ar);
        const float4 a1 = *reinterpret_cast<const float4*>(ar + 8);
        a[0] = a0.x; a[1] = a0.y; a[2] = a0.z; a[3] = a0.w;
        a[4] = a1.x; a[5] = a1.y; a[6] = a1.z; a[7] = a1.w;
      } else {
#pragma unroll
        for (int st = 0; st < 8; ++st)
          a[st] = As[buf][(8 * (st >> 2) + 4 * lhi + (st & 3)) * G2_BM + wave * 32 + l31];
      }
      const float* bbase = &Bs[buf][(4 * lhi) * BN];
      float4 b0[2];
      float2 b1[2];
      float b2[2];
      auto bread = [&](int st, int slot) {
        const float* br = bbase + (8 * (st >> 2) + (st & 3)) * BN;
        b0[slot] = *reinterpret_cast<const float4*>(br + 4 * l31);
        if (NF == 7) {
          b1[slot] = *reinterpret_cast<const float2*>(br + 128 + 2 * l31);
          b2[slot] = br[192 + l31];
        }
      };
      bread(0, 0);
#pragma unroll
      for (int st = 0; st < 8; ++st) {
        const int cur = st & 1;
        if (st + 1 < 8) bread(st + 1, cur ^ 1);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[st], b0[cur].x, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[st], b0[cur].y, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[st], b0[cur].z, acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[st], b0[cur].w, acc[3], 0, 0, 0);
        if (NF == 7) {
          acc[4] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[st], b1[cur].x, acc[4], 0, 0, 0);
          acc[5] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[st], b1[cur].y, acc[5], 0, 0, 0);
          acc[6] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[st], b2[cur], acc[6], 0, 0, 0);
        }
      }
    };
    if (nfast > 0) fast_load(); else edge_load(kbeg);
    stage_store(0);
    __syncthreads();
    int kt = 0;
    for (; kt + 1 < nfast; ++kt) {             // steady state: the next chunk is interior too
      fast_load();
      compute(kt & 1);
      stage_store((kt & 1) ^ 1);
      __syncthreads();
    }
    for (; kt < nk; ++kt) {                    // K tail / edge tiles
      if (kt + 1 < nk) edge_load(kbeg + (kt + 1) * G2_BK);
      compute(kt & 1);
      if (kt + 1 < nk) stage_store((kt & 1) ^ 1);
      __syncthreads();
    }

    // epilogue: acc[j][r] -> row m0 + wave*32 + (r&3) + 8*(r>>2) + 4*lhi ; column by fragment.
    // ks == 1: final values go straight to C (plain stores, or += when accumulating);
    // ks  > 1: this item's partial sums go to its slab part[ksp][row][col] (plain stores) and
    //          gemm_wide_reduce_kernel adds the slabs in a fixed order.
    const bool direct = (ks == 1);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
      if (row >= Mlim) continue;
      float* c;
      if (direct) {
        const long crow = (p.dev_bound == 1) ? (long)p.rows[p.r0 + row] : (long)row;
        c = p.C + crow * p.ldc + n0;
      } else {
        c = p.part + ((long)(ksp * mt + mtile) * G2_BM + (row - m0)) * p.part_ld + n0;
      }
      const bool acc_c = direct && p.beta;
      auto put = [&](int cj, float v) {            // one element: a real column, the ones column, or nothing
        if (cj < ncols) c[cj] = v + (acc_c ? c[cj] : 0.f);
        else if (cj == ones_col && ntile == 0) {
          if (direct) p.extra[row] = v + (p.beta ? p.extra[row] : 0.f);
          else c[p.N] = v;                         // slab column N (n-tile 0: n0 == 0)
        }
      };
      // fragments 0-3: four consecutive columns per lane
      const int c0 = 4 * l31;
      if (c0 + 3 < ncols && (direct ? p.vecC : 1)) {
        float4 v = make_float4(acc[0][r], acc[1][r], acc[2][r], acc[3][r]);
        float4* d = reinterpret_cast<float4*>(c + c0);
        if (acc_c) { const float4 o = *d; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
        *d = v;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) put(c0 + j, acc[j][r]);
      }
      if (NF == 7) {
        put(128 + 2 * l31, acc[4][r]);
        put(128 + 2 * l31 + 1, acc[5][r]);
        put(192 + l31, acc[6][r]);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// The same products on the bf16 matrix pipe with error compensation (see ce_fast.hip), here with THREE bf16 terms per
// fp32 operand (hi + mid + lo = 24 significant bits, the whole fp32 value) and the six products whose weight is above
// 2^-24 (hi.hi, hi.mid, mid.hi, hi.lo, lo.hi, mid.mid): "bf16x6".  These are the gradient products of the training
// step, which feed long sums with cancellation (a decoder's d init is a sum over ~200 timesteps): with two terms
// ("bf16x3", relative error ~2^-17 per product) the worst gradient of the reference fixtures moved from 2.4e-5 to
// 2.5e-4 of its tensor's largest entry -- inside the 1e-3 bar but ten times the fp32 kernels' error -- while the
// kernel is bound by its staging, not by the matrix pipe, so the three extra passes cost little.  Work layout,
// k-ranges, live-row lists, ones column, slabs and reduction are exactly those of gemm_wide_kernel; what differs is
// the inner machinery.  v_mfma_f32_32x32x16_bf16 wants, per lane, 8 CONSECUTIVE k of one row of A and of one column of B,
// while B (and A in the weight-gradient products) is k-major in memory.  So the staging pass transposes: a thread
// loads an 8(k) x 4(columns) block with eight 16-byte loads, splits every value into its bf16 terms and writes, per
// column, the 8 k-values as one 16-byte slot of a [column][16 k] LDS image (one image per term).  A k-contiguous
// A needs no transpose (two 16-byte loads = one slot).  Rows of the images are 32 bytes (two slots); the slot index
// is XORed with (row >> 3) & 1, which makes the b128 fragment reads of 16 consecutive rows conflict-free.
// Fragment j of a wave covers columns 32j .. 32j+31 (natural order: the epilogue stores one column per lane).
// Requires 16-byte aligned operands with leading dimensions that are multiples of 4 (the launcher checks).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
#ifndef GW_DEEP_RING
#define GW_DEEP_RING 0      // 1: two-chunk-ahead register ring (measured: spills at 2 waves per SIMD)
#endif

// GW_EXP_* : timing-only diagnostic builds (scripts/build_variant.sh); results are wrong with any of them set
#ifndef GW_EXP_NOSPLIT
#define GW_EXP_NOSPLIT 0
#endif
#ifndef GW_EXP_NOMFMA
#define GW_EXP_NOMFMA 0
#endif
#ifndef GW_EXP_NOLOAD
#define GW_EXP_NOLOAD 0
#endif
#ifndef GW_EXP_NOLDS
#define GW_EXP_NOLDS 0
#endif
#ifndef GW_EXP_NOB
#define GW_EXP_NOB 0       // B tile fetched, split and stored for the first two chunks of an item only
#endif
__device__ __forceinline__ void gw_split(const f32x8 v, bf16x8 (&t)[3]) {
#if GW_EXP_NOSPLIT
  union { f32x8 f; bf16x8 h[2]; } u;
  u.f = v;
  t[0] = u.h[0]; t[1] = u.h[1]; t[2] = u.h[0];
#else
  t[0] = __builtin_convertvector(v, bf16x8);
  const f32x8 r1 = v - __builtin_convertvector(t[0], f32x8);
  t[1] = __builtin_convertvector(r1, bf16x8);
  t[2] = __builtin_convertvector(r1 - __builtin_convertvector(t[1], f32x8), bf16x8);
#endif
}

template <bool AKC, int NF>
__global__ __launch_bounds__(256, 2) void gemm_wide_bf3_kernel(G224Batch batch) {
  const G224Args& p = batch.a[blockIdx.y];       // one product per grid row (launch_gemm_wide_n)
  constexpr int BN = NF * 32;
  // LDS, in 16-byte slots: [buf][A hi: 128 rows x 2][A mid][A lo][B hi: BN rows x 2][B mid][B lo]
  constexpr int A_SL = G2_BM * 2, B_SL = BN * 2, BUF_SL = 3 * A_SL + 3 * B_SL;
  __shared__ __attribute__((aligned(16))) bf16x8 sm[2 * BUF_SL];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const WideLayout L = wide_layout(p, gridDim.x);
  const int Mlim = L.Mlim, Klim = L.Klim;
  if (Mlim <= 0 || Klim <= 0) return;
  const int mt = L.mt, nt = p.ntiles, ks = L.ks, kchunk = L.kchunk;
  const int items = mt * nt * ks;
  const int tw = p.tile_w;
  const int nf4 = tw >> 2;                     // 4-column groups of a B tile
  const int ones_col = p.extra ? tw : -1;

  // image rows that staging never writes (columns >= tw): zero, except the ones column (1.0 = hi 1, lo 0)
  {
    bf16x8 zero, one;
#pragma unroll
    for (int j = 0; j < 8; ++j) { zero[j] = (__bf16)0.f; one[j] = (__bf16)1.f; }     // 1.0 = hi 1, mid 0, lo 0
    for (int i = tid; i < 2 * 3 * B_SL; i += 256) {
      const int buf = i / (3 * B_SL), r = i % (3 * B_SL), img = r / B_SL, row = (r % B_SL) >> 1;
      if (row >= tw) sm[buf * BUF_SL + 3 * A_SL + img * B_SL + (r % B_SL)] = (row == ones_col && img == 0) ? one : zero;
    }
  }
  // staging roles.  A k-contiguous: every thread one (row, k-half) slot.  A k-major: threads 0..63 one
  // (4 columns, k-half) block.  B: nb_tasks = 2 * nf4 blocks on threads tb0 .. tb0 + nb_tasks - 1.
  const int tb0 = AKC ? 0 : 64;
  const bool has_b = tid >= tb0 && tid < tb0 + 2 * nf4;
  const int b_cg = has_b ? (tid - tb0) % nf4 : 0, b_kh = has_b ? (tid - tb0) / nf4 : 0;
  const bool has_a = AKC || tid < 64;
  const int a_row = tid >> 1, a_kh = AKC ? (tid & 1) : (tid >> 5), a_cg = tid & 31;
  __syncthreads();

  for (int item = blockIdx.x; item < items; item += gridDim.x) {
    const int ntile = item % nt, mtile = (item / nt) % mt, ksp = item / (nt * mt);
    const int m0 = mtile * G2_BM, n0 = ntile * tw;
    const int ncols = min(tw, p.N - n0);
    const int kbeg = ksp * kchunk, kend = min(Klim, kbeg + kchunk);

    // Staging registers, two sets (chunks are fetched TWO chunks ahead of their MFMAs: a chunk of three bf16 passes
    // is shorter than an L2 round trip).  A k-major A block and a B block never share a thread, so one 8 x float4
    // block per set serves both roles; a k-contiguous A adds its two float4 per set.  The deep ring is dropped
    // where the registers do not allow it (k-contiguous A with 224-wide tiles).
    constexpr bool DEEP = (GW_DEEP_RING != 0) && !(AKC && NF == 7);
    float4 ra_[2][AKC ? 2 : 1], rt_[2][8];
    // one K chunk (16 k from k0) of this item into registers; out-of-range k / rows / columns read as zero
    auto load = [&](int k0, float4 (&ra)[AKC ? 2 : 1], float4 (&rt)[8]) {
      if (AKC) {
        const int mm = min(m0 + a_row, Mlim - 1);
        const long arow = p.a_ind ? (long)p.rows[p.r0 + mm] : (long)mm;
        const int k = k0 + 8 * a_kh;
        const float* src = p.A + arow * p.lda + k;
        ra[0] = (k + 3 < kend) ? *reinterpret_cast<const float4*>(src) : ld4g(src, k, kend, 0);
        ra[1] = (k + 7 < kend) ? *reinterpret_cast<const float4*>(src + 4) : ld4g(src + 4, k + 4, kend, 0);
      } else if (has_a) {
        const int m = m0 + 4 * a_cg;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
          const int k = k0 + 8 * a_kh + kk;
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
          if (k < kend && m < Mlim) {
            const long arow = p.a_ind ? (long)p.rows[p.r0 + k] : (long)k;
            const float* src = p.A + arow * p.lda + m;
            v = (m + 3 < Mlim) ? *reinterpret_cast<const float4*>(src) : ld4g(src, m, Mlim, 0);
          }
          rt[kk] = v;
        }
      }
      if (has_b) {
        const int c = 4 * b_cg;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
          const int k = k0 + 8 * b_kh + kk;
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
          if (k < kend && c < ncols) {
            const long brow = (p.dev_bound == 2) ? (long)p.rows[p.r0 + k] : (long)k;
            const float* src = p.B + brow * p.ldb + n0 + c;
            v = (c + 3 < ncols) ? *reinterpret_cast<const float4*>(src) : ld4g(src, c, ncols, 0);
          }
          rt[kk] = v;
        }
      }
    };
    bool do_b = true;      // (GW_EXP_NOB)
    auto store = [&](int buf, const float4 (&ra)[AKC ? 2 : 1], const float4 (&rt)[8]) {
      bf16x8* base = sm + buf * BUF_SL;
      if (AKC) {
        f32x8 v;
        v[0] = ra[0].x; v[1] = ra[0].y; v[2] = ra[0].z; v[3] = ra[0].w;
        v[4] = ra[1].x; v[5] = ra[1].y; v[6] = ra[1].z; v[7] = ra[1].w;
        bf16x8 t[3];
        gw_split(v, t);
        const int sl = a_row * 2 + (a_kh ^ ((a_row >> 3) & 1));
#pragma unroll
        for (int q = 0; q < 3; ++q) base[q * A_SL + sl] = t[q];
      } else if (has_a) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          f32x8 v;
#pragma unroll
          for (int kk = 0; kk < 8; ++kk) v[kk] = reinterpret_cast<const float*>(&rt[kk])[j];
          bf16x8 t[3];
          gw_split(v, t);
          const int row = 4 * a_cg + j, sl = row * 2 + (a_kh ^ ((row >> 3) & 1));
#pragma unroll
          for (int q = 0; q < 3; ++q) base[q * A_SL + sl] = t[q];
        }
      }
      if (has_b && do_b) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          f32x8 v;
#pragma unroll
          for (int kk = 0; kk < 8; ++kk) v[kk] = reinterpret_cast<const float*>(&rt[kk])[j];
          bf16x8 t[3];
          gw_split(v, t);
          const int row = 4 * b_cg + j, sl = row * 2 + (b_kh ^ ((row >> 3) & 1));
#pragma unroll
          for (int q = 0; q < 3; ++q) base[3 * A_SL + q * B_SL + sl] = t[q];
        }
      }
    };

    f32x16 acc[NF];
#pragma unroll
    for (int j = 0; j < NF; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    const int arow_l = wave * 32 + l31;
    const int a_sl = arow_l * 2 + (lhi ^ ((arow_l >> 3) & 1));
    const int b_sw = lhi ^ ((l31 >> 3) & 1);            // rows 32j + l31: (row >> 3) & 1 == (l31 >> 3) & 1
    auto compute = [&](int buf) {
      const bf16x8* base = sm + buf * BUF_SL;
      const bf16x8 ah = base[a_sl], am = base[A_SL + a_sl], al = base[2 * A_SL + a_sl];
#pragma unroll
      for (int j = 0; j < NF; ++j) {
        const int bs = 3 * A_SL + (32 * j + l31) * 2 + b_sw;
        const bf16x8 bh = base[bs], bm = base[B_SL + bs], bl = base[2 * B_SL + bs];
#if GW_EXP_NOMFMA
        acc[j][0] += (float)am[0] * (float)bm[0] + (float)ah[1] * (float)bl[1] + (float)al[2] * (float)bh[2];
#else
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc[j], 0, 0, 0);      // smallest terms first
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[j], 0, 0, 0);
#endif
      }
    };

    // interior chunks (every load a whole in-bounds float4: full k range, full column groups, and for a k-major A
    // a full row tile) take a branch-free path on per-task base pointers; k-major operands that go through the
    // row list fetch their 8 row indices per chunk with two 16-byte loads
    // (A k-major row tile that sticks out past Mlim: its column groups are clamped to the last whole one -- the
    // duplicated columns only feed accumulator rows that are never stored.  Left on the guarded path, the items of
    // the LAST row tile ran several times longer than every other item and set the launch's duration: the
    // [dW | db] product of ToVocab, 157 row tiles, 353 -> see DESIGN 4.)
    const bool fast_item = ncols == tw && (AKC || m0 + G2_BM <= Mlim || Mlim >= 4);
    // the one column group that straddles Mlim (Mlim not a multiple of 4) is fetched element by element
    const int a_m = m0 + 4 * a_cg;
    const bool a_straddle = !AKC && has_a && a_m < Mlim && a_m + 4 > Mlim;
    const float* fa;
    const float* fb = p.B + n0 + 4 * b_cg;
    if (AKC) {
      const int mm = min(m0 + a_row, Mlim - 1);
      fa = p.A + (p.a_ind ? (long)p.rows[p.r0 + mm] : (long)mm) * p.lda + 8 * a_kh;
    } else {
      fa = p.A + (a_m + 4 <= Mlim ? a_m : (a_straddle ? a_m : ((Mlim - 4) & ~3)));
    }
    const bool rows16 = ((p.r0 & 3) == 0) && ((reinterpret_cast<uintptr_t>(p.rows) & 15) == 0);   // 16-byte aligned slices of 4 indices
    auto fast_load = [&](int k0, float4 (&ra)[AKC ? 2 : 1], float4 (&rt)[8]) {
      if (AKC) {
        ra[0] = *reinterpret_cast<const float4*>(fa + k0);
        ra[1] = *reinterpret_cast<const float4*>(fa + k0 + 4);
      } else if (has_a) {
        const int kb = k0 + 8 * a_kh;
        if (p.a_ind) {
          int ri[8];
          if (rows16) {
            const int4 q0 = *reinterpret_cast<const int4*>(p.rows + p.r0 + kb), q1 = *reinterpret_cast<const int4*>(p.rows + p.r0 + kb + 4);
            ri[0] = q0.x; ri[1] = q0.y; ri[2] = q0.z; ri[3] = q0.w; ri[4] = q1.x; ri[5] = q1.y; ri[6] = q1.z; ri[7] = q1.w;
          } else {
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) ri[kk] = p.rows[p.r0 + kb + kk];
          }
          if (a_straddle) {
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) rt[kk] = ld4g(fa + (long)ri[kk] * p.lda, a_m, Mlim, 0);
          } else {
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) rt[kk] = *reinterpret_cast<const float4*>(fa + (long)ri[kk] * p.lda);
          }
        } else if (a_straddle) {
#pragma unroll
          for (int kk = 0; kk < 8; ++kk) rt[kk] = ld4g(fa + (long)(kb + kk) * p.lda, a_m, Mlim, 0);
        } else {
#pragma unroll
          for (int kk = 0; kk < 8; ++kk) rt[kk] = *reinterpret_cast<const float4*>(fa + (long)(kb + kk) * p.lda);
        }
      }
      if (has_b && do_b) {
        const int kb = k0 + 8 * b_kh;
        if (p.dev_bound == 2) {
          int ri[8];
          if (rows16) {
            const int4 q0 = *reinterpret_cast<const int4*>(p.rows + p.r0 + kb), q1 = *reinterpret_cast<const int4*>(p.rows + p.r0 + kb + 4);
            ri[0] = q0.x; ri[1] = q0.y; ri[2] = q0.z; ri[3] = q0.w; ri[4] = q1.x; ri[5] = q1.y; ri[6] = q1.z; ri[7] = q1.w;
          } else {
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) ri[kk] = p.rows[p.r0 + kb + kk];
          }
#pragma unroll
          for (int kk = 0; kk < 8; ++kk) rt[kk] = *reinterpret_cast<const float4*>(fb + (long)ri[kk] * p.ldb);
        } else {
#pragma unroll
          for (int kk = 0; kk < 8; ++kk) rt[kk] = *reinterpret_cast<const float4*>(fb + (long)(kb + kk) * p.ldb);
        }
      }
    };
    const int nk = (kend - kbeg + G2_BK - 1) / G2_BK;
    const int nfast = fast_item ? (kend - kbeg) / G2_BK : 0;   // interior chunks form a prefix
    auto fetch = [&](int c, float4 (&ra)[AKC ? 2 : 1], float4 (&rt)[8]) {      // chunk c of the item -> a register set
#if GW_EXP_NOLOAD
      if (c > 0 && c < nfast) return;
#endif
      if (c < nfast) fast_load(kbeg + c * G2_BK, ra, rt);
      else if (c < nk) load(kbeg + c * G2_BK, ra, rt);
    };
    if (DEEP) {
      fetch(0, ra_[0], rt_[0]);
      store(0, ra_[0], rt_[0]);
      fetch(1, ra_[1], rt_[1]);
      __syncthreads();
      for (int kt = 0; kt < nk; kt += 2) {
        // chunk kt in LDS buffer 0, chunk kt+1 in register set 1
        fetch(kt + 2, ra_[0], rt_[0]);
        compute(0);
        if (kt + 1 < nk) store(1, ra_[1], rt_[1]);
        __syncthreads();
        if (kt + 1 >= nk) break;
        // chunk kt+1 in LDS buffer 1, chunk kt+2 in register set 0
        fetch(kt + 3, ra_[1], rt_[1]);
        compute(1);
        if (kt + 2 < nk) store(0, ra_[0], rt_[0]);
        __syncthreads();
      }
    } else {
      fetch(0, ra_[0], rt_[0]);
      store(0, ra_[0], rt_[0]);
      __syncthreads();
      for (int kt = 0; kt < nk; ++kt) {
#if GW_EXP_NOB
        do_b = kt < 1;
#endif
        fetch(kt + 1, ra_[0], rt_[0]);
        compute(kt & 1);
#if GW_EXP_NOLDS
        if (kt + 1 < nk && kt < 1) store((kt & 1) ^ 1, ra_[0], rt_[0]);
#else
        if (kt + 1 < nk) store((kt & 1) ^ 1, ra_[0], rt_[0]);
#endif
        __syncthreads();
      }
    }

    // epilogue: acc[j][r] -> row m0 + wave*32 + (r&3) + 8*(r>>2) + 4*lhi, column 32j + l31 of the tile
    const bool direct = (ks == 1);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
      if (row >= Mlim) continue;
      float* c;
      if (direct) {
        const long crow = (p.dev_bound == 1) ? (long)p.rows[p.r0 + row] : (long)row;
        c = p.C + crow * p.ldc + n0;
      } else {
        c = p.part + ((long)(ksp * mt + mtile) * G2_BM + (row - m0)) * p.part_ld + n0;
      }
      const bool acc_c = direct && p.beta;
#pragma unroll
      for (int j = 0; j < NF; ++j) {
        const int cj = 32 * j + l31;
        const float v = acc[j][r];
        if (cj < ncols) c[cj] = v + (acc_c ? c[cj] : 0.f);
        else if (cj == ones_col && ntile == 0) {
          if (direct) p.extra[row] = v + (p.beta ? p.extra[row] : 0.f);
          else c[p.N] = v;
        }
      }
    }
  }
}

// C (+)= sum over the ks slabs, in slab order (deterministic).  One thread per (row, 4 columns).
__global__ __launch_bounds__(256) void gemm_wide_reduce_kernel(G224Batch batch, int nblk) {
  const G224Args& p = batch.a[blockIdx.y];
  const WideLayout L = wide_layout(p, nblk);
  if (L.ks <= 1 || L.Mlim <= 0 || L.Klim <= 0) return;
  const int ncol = p.N + (p.extra ? 1 : 0);
  const int nc4 = (ncol + 3) >> 2;
  const long total = (long)L.Mlim * nc4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int row = (int)(i / nc4), c0 = 4 * (int)(i % nc4);
    float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = 0; k < L.ks; ++k) {
      const float4 v = *reinterpret_cast<const float4*>(
          p.part + ((long)(k * L.mt + row / G2_BM) * G2_BM + row % G2_BM) * p.part_ld + c0);
      sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
    }
    const long crow = (p.dev_bound == 1) ? (long)p.rows[p.r0 + row] : (long)row;
    float* c = p.C + crow * p.ldc;
    const float sv[4] = {sum.x, sum.y, sum.z, sum.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = c0 + j;
      if (col < p.N) c[col] = sv[j] + (p.beta ? c[col] : 0.f);
      else if (p.extra && col == p.N) p.extra[row] = sv[j] + (p.beta ? p.extra[row] : 0.f);
    }
  }
}

// Partial-sum slabs: at most persistent_grid_blocks() tiles of 128 rows are in flight per launch.
long gemm_wide_part_floats(int N, bool extra) {
  const long ld = (N + (extra ? 1 : 0) + 3) / 4 * 4;
  return (long)wide_grid_blocks() * G2_BM * ld;
}

// Library-owned scratch for callers that bring none: one buffer per stream (lib_scratch).
static int lib_part(long floats, float** out, hipStream_t s) {
  void* p = nullptr;
  CVAE_TRY(lib_scratch(LS_WIDE_PART, s, (size_t)floats * sizeof(float), &p));
  *out = static_cast<float*>(p);
  return 0;
}

// The ones column (bias gradient / row sums as an extra output column) needs the 224-wide tiling with a balanced
// tile width of at most 220: taken whenever such a tiling exists (a few idle MFMA columns cost less than the separate
// pass over A that the caller would launch instead), and for every N whose 224-tiling is the cheaper one anyway.
static void wide_tiling(int N, bool extra, bool* wide, int* ntiles, int* tile_w) {
  const int n_eff = N + (extra ? 1 : 0);
  const int nt7 = cdiv(n_eff, 224), nt4 = cdiv(n_eff, 128);
  const int tw7 = cdiv(cdiv(N, nt7), 4) * 4;
  *wide = (7 * nt7 < 4 * nt4) || (extra && tw7 <= 220);
  *ntiles = *wide ? nt7 : nt4;
  *tile_w = cdiv(cdiv(N, *ntiles), 4) * 4;
}
bool gemm_wide_has_spare_column(int N) {
  bool wide; int nt, tw;
  wide_tiling(N, true, &wide, &nt, &tw);
  return wide && tw <= 220;
}

// n products of one shape (same M, N, K bounds, operand forms and options; different operands) in ONE launch: grid row y
// runs product y on its share of the persistent grid.  The small products of a training step are bound by the launch
// itself and by a chain of memory round trips (device-side bound, row list, first chunks), not by their arithmetic: the
// two decoders' (or the two encoder directions') products of one kind cost one such chain instead of two.
int launch_gemm_wide_n(bool akc, const G224Args* in, int n, hipStream_t s) {
  CVAE_REQUIRE(in && n >= 1 && n <= G224_MAX_BATCH, "gemm_wide: %d products in one launch (max %d)", n, G224_MAX_BATCH);
  G224Batch batch = {};
  G224Args& a = batch.a[0];
  a = in[0];
  CVAE_REQUIRE(a.N > 0, "gemm_wide: N=%d", a.N);
  CVAE_REQUIRE(a.extra == nullptr || gemm_wide_has_spare_column(a.N), "gemm_wide: no spare tile column for N=%d", a.N);
  for (int i = 0; i < n; ++i) {
    const G224Args& q = in[i];
    CVAE_REQUIRE(q.M == a.M && q.N == a.N && q.K == a.K && q.dev_bound == a.dev_bound && q.a_ind == a.a_ind &&
                 (q.extra != nullptr) == (a.extra != nullptr) && q.beta == a.beta && (q.part != nullptr) == (a.part != nullptr),
                 "gemm_wide: products of one launch differ in shape or form");
    CVAE_REQUIRE(!q.a_ind || (q.rows && ((akc && q.dev_bound == 1) || (!akc && q.dev_bound == 2))),
                 "gemm_wide: A-row indirection needs the row list and the matching device bound");
    CVAE_REQUIRE(n == 1 || !q.part, "gemm_wide: a batched launch takes its slabs from the library");
  }
  if (a.M <= 0 || a.K <= 0) return 0;
  // balanced n-tiles: 224-wide fragments when that saves MFMA columns, else 128-wide
  const int n_eff = a.N + (a.extra ? 1 : 0);
  const int nblk = wide_grid_blocks() / n;            // blocks per product
  bool wide;
  wide_tiling(a.N, a.extra != nullptr, &wide, &a.ntiles, &a.tile_w);
  if (!a.extra) {
    // more tiles than resident blocks (one k-range per tile): the launch runs in rounds, so pick the tile width
    // by rounds x columns per tile (8192 x 2048 outputs: 640 tiles of 224 = 2 rounds of 7, 1024 tiles of 128 = 2 of 4)
    const long mt = cdiv(a.M, G2_BM), nb = nblk;
    const long nt7 = cdiv(a.N, 224), nt4 = cdiv(a.N, 128);
    if (mt * nt7 > nb || mt * nt4 > nb) {
      const double c7 = (double)cdiv(mt * nt7, nb) * 7.0, c4 = (double)cdiv(mt * nt4, nb) * 4.0 * 1.08;
      wide = c7 <= c4;
      a.ntiles = (int)(wide ? nt7 : nt4);
      a.tile_w = cdiv(cdiv(a.N, a.ntiles), 4) * 4;
    }
  }
  a.part_ld = (n_eff + 3) / 4 * 4;
  {
    static int mc = 0;
    if (!mc) { const char* e = getenv("CVAE_WIDE_MIN_CHUNKS"); mc = e ? atoi(e) : 4; if (mc < 1 || mc > 256) mc = 4; }
    a.min_chunks = mc;
  }
  // slabs are only touched when the device picks more than one k-range per tile (same test as the reduction below)
  const bool may_split = (long)cdiv(a.M, G2_BM) * a.ntiles * 2 <= nblk || a.dev_bound;
  float* part = a.part;
  if (!part && may_split) CVAE_TRY(lib_part(gemm_wide_part_floats(a.N, a.extra != nullptr), &part, s));
  bool vec = true;
  double work = 0.0;
  for (int i = 0; i < n; ++i) {
    G224Args& q = batch.a[i];
    if (i) {
      q = in[i];
      q.ntiles = a.ntiles; q.tile_w = a.tile_w; q.part_ld = a.part_ld; q.min_chunks = a.min_chunks;
    }
    q.vecA = aligned16(q.A) && (q.lda % 4 == 0);
    q.vecB = aligned16(q.B) && (q.ldb % 4 == 0);
    q.vecC = aligned16(q.C) && (q.ldc % 4 == 0);
    vec = vec && q.vecA && q.vecB;
    // every product owns nblk tiles' worth of the slab buffer (at most nblk (tile, k-range) items have slabs)
    q.part = part ? part + (long)i * nblk * G2_BM * a.part_ld : nullptr;
    double pm = q.M, pk = q.K;
    if (q.dev_bound == 1 && q.prof_units > 0) pm = (double)q.prof_units;
    if (q.dev_bound == 2 && q.prof_units > 0) pk = (double)q.prof_units;
    work += 2.0 * pm * q.N * pk;
  }
  const dim3 grid(nblk, n), block(256);
  {
    ProfScope prof(PK_GEMM_WIDE, s, work);
    if (split_bf16_enabled() && vec) {
      if (akc) {
        if (wide) hipLaunchKernelGGL((gemm_wide_bf3_kernel<true, 7>), grid, block, 0, s, batch);
        else hipLaunchKernelGGL((gemm_wide_bf3_kernel<true, 4>), grid, block, 0, s, batch);
      } else {
        if (wide) hipLaunchKernelGGL((gemm_wide_bf3_kernel<false, 7>), grid, block, 0, s, batch);
        else hipLaunchKernelGGL((gemm_wide_bf3_kernel<false, 4>), grid, block, 0, s, batch);
      }
    } else if (akc) {
      if (wide) hipLaunchKernelGGL((gemm_wide_kernel<true, 7>), grid, block, 0, s, batch);
      else hipLaunchKernelGGL((gemm_wide_kernel<true, 4>), grid, block, 0, s, batch);
    } else {
      if (wide) hipLaunchKernelGGL((gemm_wide_kernel<false, 7>), grid, block, 0, s, batch);
      else hipLaunchKernelGGL((gemm_wide_kernel<false, 4>), grid, block, 0, s, batch);
    }
  }
  CVAE_LAUNCH_CHECK("gemm_wide_kernel");
  // the reduction is a no-op (early exit) when the device picked one k-range per tile
  if (may_split) {
    const long total = (long)a.M * (a.part_ld / 4);
    int rb = (int)((total + 255) / 256);
    if (rb > 2048 / n) rb = 2048 / n;
    hipLaunchKernelGGL(gemm_wide_reduce_kernel, dim3(rb, n), dim3(256), 0, s, batch, nblk);
    CVAE_LAUNCH_CHECK("gemm_wide_reduce_kernel");
  }
  return 0;
}

int launch_gemm_wide(bool akc, G224Args a, hipStream_t s) { return launch_gemm_wide_n(akc, &a, 1, s); }

}  // namespace cvae
