// One recurrent step for many rows (importance-weighted evaluation, R = samples x sentences):
//   h_t = cell(h_{t-1} W_hh^T + b_hh, x-side pre-activations, h_{t-1} [, c_{t-1}])
// as ONE kernel: the gate product on the matrix cores with the GRU / LSTM cell in its epilogue.
// The unfused form (tiled GEMM into hp[R, G*H], then a pointwise kernel) writes and re-reads
// G*H floats per row and step -- 262 MB per step at R = 16000, H = 512 -- which costs as much as
// the product itself.
//
//  * block tile: 128 rows x 32 hidden units x all G gates.  W_hh rows g*H + u0 + (0..31) of the
//    G gates are staged as G groups of 32 LDS rows, so a wave's 32x32 MFMA block j IS gate j of
//    its 32 units and a lane ends up with all G pre-activations of (row, unit): the cell is
//    lane-local, no shuffles, no LDS round trip;
//  * both operands are k-contiguous ([row][k]); LDS rows of BK + 4 floats make the b128 fragment
//    reads conflict-free; lane (l31, lhi) reads k = 8q + 4*lhi + (0..3) of its row for A and B
//    alike, i.e. both sides see the same permutation of k (v_mfma_f32_32x32x2_f32, 4 steps per read);
//  * double-buffered LDS, one barrier per RL_BK-wide K chunk (16: 354 us per 16 000-row step against 375 us
//    with 32), every global load a whole float4
//    (rows past the live prefix are clamped to a valid row; their results are not stored).
#include "common.hpp"

namespace cvae {

#ifndef RL_BK_
#define RL_BK_ 16
#endif
#ifndef RL_OCC_
#define RL_OCC_ 2
#endif
constexpr int RL_BM = 128, RL_BU = 32, RL_BK = RL_BK_, RL_LD = RL_BK + 4, RL_F4 = RL_BK / 4;   // RL_LD / 4 odd: conflict-free b128

struct RnnLargeArgs {
  const float* h_prev; const float* c_prev; const float* W_hh; const float* b_hh;
  const float* xproj; const float* xadd;      // xproj row (r / xdiv) % xmod, xadd row r (or null)
  float* h_out; float* c_out;
  int live, H, xmod, xdiv;
};

template <int CELL>
__global__ __launch_bounds__(256, RL_OCC_) void rnn_large_step_kernel(RnnLargeArgs p) {
  constexpr int G = CELL == CELL_GRU ? 3 : 4;
  extern __shared__ __attribute__((aligned(16))) float rl_smem[];       // 2 x (A tile + B tile) = 72 KB (LSTM)
  float (*As)[RL_BM * RL_LD] = reinterpret_cast<float (*)[RL_BM * RL_LD]>(rl_smem);
  float (*Bs)[G * RL_BU * RL_LD] = reinterpret_cast<float (*)[G * RL_BU * RL_LD]>(rl_smem + 2 * RL_BM * RL_LD);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int H = p.H, GH = G * H;
  // XCD-aware tile order: workgroups go round-robin over the 8 XCDs, each with its own L2.  All unit
  // tiles of one 128-row tile are given to one XCD (its h rows are then fetched into that L2 once).
  const int nu = H / RL_BU, L = blockIdx.x;
  const int rowtile = ((L >> 3) / nu) * 8 + (L & 7), utile = (L >> 3) % nu;
  const int u0 = utile * RL_BU, m0 = rowtile * RL_BM;
  if (m0 >= p.live) return;

  // staging: A 128 rows x RL_F4 float4, B (G*32) rows x RL_F4 float4; thread -> (row = idx / RL_F4, k4 = idx % RL_F4)
  constexpr int NA = RL_BM * (RL_BK / 4) / 256;          // 4
  constexpr int NBT = G * RL_BU * (RL_BK / 4);           // float4 pieces of the B tile: 384 (GRU) / 512 (LSTM)
  constexpr int NB = (NBT + 255) / 256;                  // 2; the GRU's second pass covers half the block
  constexpr bool NB_EXACT = NB * 256 == NBT;
  // element offsets from the kernel-argument base pointers (kept as offsets so that the loads stay
  // global_load: pointer arrays that are advanced in the loop degrade to flat loads + scratch)
  const float* __restrict__ gA = p.h_prev;
  const float* __restrict__ gB = p.W_hh;
  long oa[NA], ob[NB];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int idx = tid + 256 * i;
    const int row = min(m0 + idx / RL_F4, p.live - 1);
    oa[i] = (long)row * H + 4 * (idx % RL_F4);
  }
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int idx = tid + 256 * i;
    const int br = min(idx / RL_F4, G * RL_BU - 1);       // 0 .. G*32-1: gate br / 32, unit br % 32 (GRU: the
                                                          // surplus threads of the last pass re-load the last row)
    ob[i] = (long)((br >> 5) * H + u0 + (br & 31)) * H + 4 * (idx % RL_F4);
  }
  int kload = 0;                                          // k of the next chunk to load
  f32x4 sa[NA], sb[NB];
#define RL_STAGE_LOAD()                                                                          \
  do {                                                                                           \
    _Pragma("unroll") for (int i = 0; i < NA; ++i) {                                             \
      sa[i] = *reinterpret_cast<const f32x4*>(gA + oa[i] + kload);                               \
    }                                                                                            \
    _Pragma("unroll") for (int i = 0; i < NB; ++i) {                                             \
      sb[i] = *reinterpret_cast<const f32x4*>(gB + ob[i] + kload);                               \
    }                                                                                            \
    kload += RL_BK;                                                                              \
  } while (0)
#define RL_STAGE_STORE(buf)                                                                      \
  do {                                                                                           \
    _Pragma("unroll") for (int i = 0; i < NA; ++i) {                                             \
      const int idx = tid + 256 * i;                                                             \
      *reinterpret_cast<f32x4*>(&As[buf][(idx / RL_F4) * RL_LD + 4 * (idx % RL_F4)]) = sa[i];           \
    }                                                                                            \
    _Pragma("unroll") for (int i = 0; i < NB; ++i) {                                             \
      const int idx = tid + 256 * i;                                                             \
      if (NB_EXACT || idx < NBT)                                                                 \
        *reinterpret_cast<f32x4*>(&Bs[buf][(idx / RL_F4) * RL_LD + 4 * (idx % RL_F4)]) = sb[i];  \
    }                                                                                            \
  } while (0)

  f32x16 acc[G];
#pragma unroll
  for (int g = 0; g < G; ++g)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;

  const int nk = H / RL_BK;
  RL_STAGE_LOAD();
  RL_STAGE_STORE(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) RL_STAGE_LOAD();
    const float* ar = &As[buf][(wave * 32 + l31) * RL_LD + 4 * lhi];
    const float* br = &Bs[buf][l31 * RL_LD + 4 * lhi];
#pragma unroll
    for (int q = 0; q < RL_BK / 8; ++q) {
      const float4 a = *reinterpret_cast<const float4*>(ar + 8 * q);
      float4 b[G];
#pragma unroll
      for (int g = 0; g < G; ++g) b[g] = *reinterpret_cast<const float4*>(br + g * RL_BU * RL_LD + 8 * q);
#pragma unroll
      for (int g = 0; g < G; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b[g].x, acc[g], 0, 0, 0);
#pragma unroll
      for (int g = 0; g < G; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b[g].y, acc[g], 0, 0, 0);
#pragma unroll
      for (int g = 0; g < G; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b[g].z, acc[g], 0, 0, 0);
#pragma unroll
      for (int g = 0; g < G; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b[g].w, acc[g], 0, 0, 0);
    }
    if (kt + 1 < nk) RL_STAGE_STORE(buf ^ 1);
    __syncthreads();
  }

  // cell: lane = unit u0 + l31; acc[g][r] = gate g of row m0 + wave*32 + (r&3) + 8*(r>>2) + 4*lhi
  const int u = u0 + l31;
  float bh[G];
#pragma unroll
  for (int g = 0; g < G; ++g) bh[g] = p.b_hh[g * H + u];
  // rows in four groups of four: all of a group's loads are issued before its arithmetic (one memory
  // round trip per group instead of one per row); rows past the live prefix load a clamped row and
  // are simply not stored.
#pragma unroll
  for (int rg = 0; rg < 4; ++rg) {
    float x[4][G], hp[4], cp[4];
    int mm[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = m0 + wave * 32 + j + 8 * rg + 4 * lhi;
      mm[j] = m;
      const int mc = min(m, p.live - 1);
      const float* xp = p.xproj + (long)((mc / p.xdiv) % p.xmod) * GH + u;
#pragma unroll
      for (int g = 0; g < G; ++g) x[j][g] = xp[g * H];
      if (p.xadd) {
        const float* xa = p.xadd + (long)mc * GH + u;
#pragma unroll
        for (int g = 0; g < G; ++g) x[j][g] += xa[g * H];
      }
      hp[j] = p.h_prev[(long)mc * H + u];
      cp[j] = (CELL == CELL_LSTM) ? p.c_prev[(long)mc * H + u] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = 4 * rg + j;
      float sg[G];
#pragma unroll
      for (int g = 0; g < G; ++g) sg[g] = acc[g][r] + bh[g];
      const long o = (long)mm[j] * H + u;
      if constexpr (CELL == CELL_GRU) {
        const float rgate = sigmoidf_(x[j][0] + sg[0]);
        const float zg = sigmoidf_(x[j][1] + sg[1]);
        const float nn = tanhf_(x[j][2] + rgate * sg[2]);
        if (mm[j] < p.live) p.h_out[o] = (1.f - zg) * nn + zg * hp[j];
      } else {
        const float ig = sigmoidf_(x[j][0] + sg[0]);
        const float fg = sigmoidf_(x[j][1] + sg[1]);
        const float gg = tanhf_(x[j][2] + sg[2]);
        const float og = sigmoidf_(x[j][3] + sg[3]);
        const float cnew = fg * cp[j] + ig * gg;
        if (mm[j] < p.live) {
          p.h_out[o] = og * tanhf_(cnew);
          p.c_out[o] = cnew;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// The same step with the gate product on the bf16 matrix pipe, error-compensated ("bf16x3", see ce_fast.hip):
// h and W_hh are split into hi + lo bf16 while they are staged (registers -> LDS), and each 16-deep K block costs
// three v_mfma_f32_32x32x16_bf16 per gate (hi.lo, lo.hi, hi.hi) instead of eight v_mfma_f32_32x32x2_f32 at twice the
// cycles each.  Tile and roles as above (128 rows x 32 units x G gates, a wave's N-tile j is gate j: the cell stays
// lane-local); K chunk 32 (two MFMA K blocks per barrier); LDS rows are 32 bf16 = four 16-byte slots, slot index
// XOR-swizzled with (row >> 2) & 3 so that the b128 fragment reads of 16 consecutive rows are conflict-free
// without padding: 2 x (A hi|lo + B hi|lo) = 64 KB, two blocks per CU.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
constexpr int RB_BK = 32;

// RB_EXP_* : timing-only diagnostic builds (scripts/build_variant.sh), results are wrong with any of them set
#ifndef RB_EXP_NOSPLIT
#define RB_EXP_NOSPLIT 0
#endif
#ifndef RB_EXP_NOEPI
#define RB_EXP_NOEPI 0
#endif
#ifndef RB_EXP_NOMFMA
#define RB_EXP_NOMFMA 0
#endif
__device__ __forceinline__ void rb_split(const f32x4 a, const f32x4 b, bf16x8& hi, bf16x8& lo) {
  f32x8 v;
  v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
#if RB_EXP_NOSPLIT
  union { f32x8 f; bf16x8 h[2]; } u;
  u.f = v;
  hi = u.h[0]; lo = u.h[1];
#else
  hi = __builtin_convertvector(v, bf16x8);
  lo = __builtin_convertvector(v - __builtin_convertvector(hi, f32x8), bf16x8);
#endif
}

template <int CELL>
__global__ __launch_bounds__(256, 2) void rnn_large_step_bf3_kernel(RnnLargeArgs p) {
  constexpr int G = CELL == CELL_GRU ? 3 : 4;
  constexpr int BROWS = G * RL_BU;                      // rows of the B tile: gate-major, 32 units each
  // LDS image, in bf16x8 (16-byte) slots: [buf][A hi: 128 rows x 4][A lo][B hi: BROWS x 4][B lo]
  constexpr int A_SL = RL_BM * 4, B_SL = BROWS * 4, BUF_SL = 2 * A_SL + 2 * B_SL;
  extern __shared__ __attribute__((aligned(16))) float rl_smem[];
  bf16x8* sm = reinterpret_cast<bf16x8*>(rl_smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int H = p.H, GH = G * H;
  const int nu = H / RL_BU, L = blockIdx.x;
  const int rowtile = ((L >> 3) / nu) * 8 + (L & 7), utile = (L >> 3) % nu;
  const int u0 = utile * RL_BU, m0 = rowtile * RL_BM;
  if (m0 >= p.live) return;

  // staging tasks: one task = 8 consecutive k of one row = one LDS slot.  A: 512 tasks, B: BROWS * 4 (384 / 512)
  constexpr int NTA = 2, NTB = (B_SL + 255) / 256;
  constexpr bool NB_EXACT = NTB * 256 == B_SL;
  const float* __restrict__ gA = p.h_prev;
  const float* __restrict__ gB = p.W_hh;
  long oa[NTA], ob[NTB];
  int sa_slot[NTA], sb_slot[NTB];
#pragma unroll
  for (int i = 0; i < NTA; ++i) {
    const int idx = tid + 256 * i, r = idx >> 2, sl = idx & 3;
    oa[i] = (long)min(m0 + r, p.live - 1) * H + 8 * sl;
    sa_slot[i] = r * 4 + (sl ^ ((r >> 2) & 3));
  }
#pragma unroll
  for (int i = 0; i < NTB; ++i) {
    const int idx = min(tid + 256 * i, B_SL - 1), r = idx >> 2, sl = idx & 3;      // r: gate r / 32, unit r % 32
    ob[i] = (long)((r >> 5) * H + u0 + (r & 31)) * H + 8 * sl;
    sb_slot[i] = r * 4 + (sl ^ ((r >> 2) & 3));
  }
  int kload = 0;
  // two register sets: the loads of chunk kt+2 are issued before chunk kt is multiplied and are split / stored to
  // LDS one iteration later (a single set left them one MFMA phase to land: the kernel ran at the latency of its
  // staging loads, not at the matrix pipe's rate)
  f32x4 ra[2][NTA][2], rb[2][NTB][2];
#define RB_STAGE_LOAD(set)                                                                       \
  do {                                                                                           \
    _Pragma("unroll") for (int i = 0; i < NTA; ++i) {                                            \
      ra[set][i][0] = *reinterpret_cast<const f32x4*>(gA + oa[i] + kload);                       \
      ra[set][i][1] = *reinterpret_cast<const f32x4*>(gA + oa[i] + kload + 4);                   \
    }                                                                                            \
    _Pragma("unroll") for (int i = 0; i < NTB; ++i) {                                            \
      rb[set][i][0] = *reinterpret_cast<const f32x4*>(gB + ob[i] + kload);                       \
      rb[set][i][1] = *reinterpret_cast<const f32x4*>(gB + ob[i] + kload + 4);                   \
    }                                                                                            \
    kload += RB_BK;                                                                              \
  } while (0)
#define RB_STAGE_STORE(set, buf)                                                                 \
  do {                                                                                           \
    bf16x8* base = sm + (buf) * BUF_SL;                                                          \
    _Pragma("unroll") for (int i = 0; i < NTA; ++i) {                                            \
      bf16x8 hi, lo;                                                                             \
      rb_split(ra[set][i][0], ra[set][i][1], hi, lo);                                            \
      base[sa_slot[i]] = hi;                                                                     \
      base[A_SL + sa_slot[i]] = lo;                                                              \
    }                                                                                            \
    _Pragma("unroll") for (int i = 0; i < NTB; ++i) {                                            \
      if (NB_EXACT || tid + 256 * i < B_SL) {                                                    \
        bf16x8 hi, lo;                                                                           \
        rb_split(rb[set][i][0], rb[set][i][1], hi, lo);                                          \
        base[2 * A_SL + sb_slot[i]] = hi;                                                        \
        base[2 * A_SL + B_SL + sb_slot[i]] = lo;                                                 \
      }                                                                                          \
    }                                                                                            \
  } while (0)

  f32x16 acc[G];
#pragma unroll
  for (int g = 0; g < G; ++g)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;

  // fragment slots of this lane: A row wave*32 + l31, B row g*32 + l31 (same swizzle term for all: (l31 >> 2) & 3)
  const int arow = wave * 32 + l31, sw = (l31 >> 2) & 3;
  const int nk = H / RB_BK;
  RB_STAGE_LOAD(0);
  RB_STAGE_STORE(0, 0);
  if (nk > 1) RB_STAGE_LOAD(1);
  __syncthreads();
  for (int kt0 = 0; kt0 < nk; kt0 += 2) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int kt = kt0 + u;
      if (kt < nk) {
        if (kt + 2 < nk) RB_STAGE_LOAD(u);                 // set u held chunk kt, already in LDS
        const bf16x8* base = sm + u * BUF_SL;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
          const int sl = (2 * kb + lhi) ^ sw;
          const bf16x8 ah = base[arow * 4 + sl], al = base[A_SL + arow * 4 + sl];
          bf16x8 bh[G], bl[G];
#pragma unroll
          for (int g = 0; g < G; ++g) {
            bh[g] = base[2 * A_SL + (g * RL_BU + l31) * 4 + sl];
            bl[g] = base[2 * A_SL + B_SL + (g * RL_BU + l31) * 4 + sl];
          }
#if RB_EXP_NOMFMA
#pragma unroll
          for (int g = 0; g < G; ++g) { acc[g][0] += (float)ah[0] * (float)bl[g][0] + (float)al[1] * (float)bh[g][1]; }
#else
#pragma unroll
          for (int g = 0; g < G; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[g], acc[g], 0, 0, 0);
#pragma unroll
          for (int g = 0; g < G; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[g], acc[g], 0, 0, 0);
#pragma unroll
          for (int g = 0; g < G; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[g], acc[g], 0, 0, 0);
#endif
        }
        if (kt + 1 < nk) RB_STAGE_STORE(u ^ 1, u ^ 1);     // chunk kt+1, loaded one iteration ago
        __syncthreads();
      }
    }
  }
#undef RB_STAGE_LOAD
#undef RB_STAGE_STORE

  // cell: identical to rnn_large_step_kernel (lane = unit u0 + l31; acc[g][r] = gate g of row
  // m0 + wave*32 + (r&3) + 8*(r>>2) + 4*lhi)
  const int u = u0 + l31;
  float bh_[G];
#pragma unroll
  for (int g = 0; g < G; ++g) bh_[g] = p.b_hh[g * H + u];
#pragma unroll
  for (int rg = 0; rg < 4; ++rg) {
    float x[4][G], hp[4], cp[4];
    int mm[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = m0 + wave * 32 + j + 8 * rg + 4 * lhi;
      mm[j] = m;
      const int mc = min(m, p.live - 1);
#if RB_EXP_NOEPI
#pragma unroll
      for (int g = 0; g < G; ++g) x[j][g] = 0.1f * g;
      hp[j] = 0.5f; cp[j] = 0.25f; (void)mc;
#else
      const float* xp = p.xproj + (long)((mc / p.xdiv) % p.xmod) * GH + u;
#pragma unroll
      for (int g = 0; g < G; ++g) x[j][g] = xp[g * H];
      if (p.xadd) {
        const float* xa = p.xadd + (long)mc * GH + u;
#pragma unroll
        for (int g = 0; g < G; ++g) x[j][g] += xa[g * H];
      }
      hp[j] = p.h_prev[(long)mc * H + u];
      cp[j] = (CELL == CELL_LSTM) ? p.c_prev[(long)mc * H + u] : 0.f;
#endif
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = 4 * rg + j;
      float sg[G];
#pragma unroll
      for (int g = 0; g < G; ++g) sg[g] = acc[g][r] + bh_[g];
      const long o = (long)mm[j] * H + u;
      if constexpr (CELL == CELL_GRU) {
        const float rgate = sigmoid_fast(x[j][0] + sg[0]);
        const float zg = sigmoid_fast(x[j][1] + sg[1]);
        const float nn = tanh_fast(x[j][2] + rgate * sg[2]);
        if (mm[j] < p.live) p.h_out[o] = (1.f - zg) * nn + zg * hp[j];
      } else {
        const float ig = sigmoid_fast(x[j][0] + sg[0]);
        const float fg = sigmoid_fast(x[j][1] + sg[1]);
        const float gg = tanh_fast(x[j][2] + sg[2]);
        const float og = sigmoid_fast(x[j][3] + sg[3]);
        const float cnew = fg * cp[j] + ig * gg;
        if (mm[j] < p.live) {
          p.h_out[o] = og * tanh_fast(cnew);
          p.c_out[o] = cnew;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Weights in registers (round 3, LSTM, H = 256 / 512).  The kernel above stages 512 KB through L2 -> LDS for every
// 128 x 128 output tile (1 GB per 16 000-row step) and runs at the rate of that staging.  Here a block keeps its
// share of W_hh for the whole launch: 8 waves (two per SIMD); wave (g, uh) holds gate g of 16 of the block's 32 units,
// all K, as split bf16 B fragments of v_mfma_f32_16x16x32_bf16 in the accumulator half of its registers (H / 32 K
// blocks x (hi, lo) x 4 = 128 AGPRs at H = 512), and the block walks over 16-row tiles of h (row group rg of nrg:
// tiles rg, rg + nrg, ...), which are the only operand that moves.
//  * h arrives already split: the cell of the previous step writes, next to its fp32 h, a bf16 (hi, lo) image of it
//    (launch_rnn_large_split makes the first one), so a tile is 16 rows x 2 x H bf16 copied global -> LDS by the LDS-DMA
//    (no registers, no conversions: every one of the H / 32 blocks that share a row group used to redo that split, and
//    the kernel ran at the rate of its vector instructions); the swizzle that makes every fragment read conflict-free
//    (slot ^ row) is on the source address.  Double-buffered: tile n+1 lands while tile n is multiplied.
//  * The H / 32 blocks that share a row group sit on one XCD, so a tile of h reaches that L2 once.
//  * The four gates of a (row, unit) end up in four waves: they cross through an LDS array X[gate][row][unit]
//    (double-buffered, one barrier per tile), and the cell runs with lane = unit, one row per thread, one tile behind
//    (inside the next tile's MFMA loop); its per-row operands (xadd, c) are requested one tile ahead.
//  * The MFMAs are inline asm: hipcc keeps MFMA A / B operands in VGPRs and would copy every weight fragment out of
//    the accumulator file before every use.  hi*lo, lo*hi and hi*hi go to three accumulators (independent MFMAs).
// WR_EXP_* : timing-only diagnostic builds (scripts/build_variant.sh), results are wrong with any of them set
#ifndef WR_EXP_NOEPILOAD
#define WR_EXP_NOEPILOAD 0
#endif
#ifndef WR_EXP_NOSTAGELOAD
#define WR_EXP_NOSTAGELOAD 0
#endif
#ifndef WR_EXP_NOMFMA
#define WR_EXP_NOMFMA 0
#endif
#ifndef WR_EXP_NOCELL
#define WR_EXP_NOCELL 0
#endif
#ifndef WR_EXP_NOB
#define WR_EXP_NOB 0
#endif
#ifndef WR_PF_
#define WR_PF_ 2
#endif
#ifndef WR_EXP_STAMPS
#define WR_EXP_STAMPS 0
#endif
#if WR_EXP_STAMPS
#define WR_STAMP(k) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); sacc_[k] += (unsigned)(now_ - tprev_); tprev_ = now_; } while (0)
#else
#define WR_STAMP(k) do {} while (0)
#endif
constexpr int WR_THREADS = 512, WR_ROWS = 16, WR_PF = WR_PF_;

struct RnnWregArgs {
  RnnLargeArgs a;
  const __bf16* img_in;      // [hi | lo][img_rows][H]: bf16 split of h_prev
  __bf16* img_out;           // the same for h_out (written by the cell)
  long img_lo;               // elements from the hi image to the lo image
  int nrg;                   // row groups
  const i32x4* wimg;         // W_hh as this kernel's B fragments: [H / 32 column groups][8 waves][K block][hi | lo][64 lanes]
};

// fp32 rows -> (hi, lo) bf16 image (the first step's h; 8 elements per thread)
__global__ __launch_bounds__(256) void rnn_large_split_kernel(long n8, const float* __restrict__ h, __bf16* __restrict__ img,
                                                              long img_lo) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n8) return;
  bf16x8 hi, lo;
  rb_split(*reinterpret_cast<const f32x4*>(h + 8 * i), *reinterpret_cast<const f32x4*>(h + 8 * i + 4), hi, lo);
  *reinterpret_cast<bf16x8*>(img + 8 * i) = hi;
  *reinterpret_cast<bf16x8*>(img + img_lo + 8 * i) = lo;
}

// W_hh -> the B fragments of rnn_large_wreg_kernel, in the order its waves load them (once per decoder call: the
// step kernel then fills its registers with 1 KB-contiguous loads instead of re-splitting its share at every step)
__global__ __launch_bounds__(WR_THREADS) void rnn_large_wimg_kernel(int H, int G, const float* __restrict__ W_hh, i32x4* __restrict__ wimg) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, kq = lane >> 4;
  const int gate = wave & 3, uh = wave >> 2, cg = blockIdx.x, nkb = H / 32;
  if (gate >= G) return;                                      // GRU: the waves of a fourth gate hold nothing
  const float* wrow = W_hh + (long)(gate * H + cg * 32 + 16 * uh + l15) * H + 8 * kq;
  i32x4* dst = wimg + (size_t)(cg * 8 + wave) * nkb * 2 * 64 + lane;
  for (int kb = 0; kb < nkb; ++kb) {
    bf16x8 hi, lo;
    rb_split(*reinterpret_cast<const f32x4*>(wrow + 32 * kb), *reinterpret_cast<const f32x4*>(wrow + 32 * kb + 4), hi, lo);
    dst[(kb * 2 + 0) * 64] = __builtin_bit_cast(i32x4, hi);
    dst[(kb * 2 + 1) * 64] = __builtin_bit_cast(i32x4, lo);
  }
}

template <int NKB, bool XADD, int CELL>                       // K blocks of 32: H = 32 NKB
__global__ __launch_bounds__(WR_THREADS) void rnn_large_wreg_kernel(RnnWregArgs w) {
  const RnnLargeArgs& p = w.a;
  constexpr int G = CELL == CELL_GRU ? 3 : 4;                 // GRU: the two waves of "gate 3" multiply nothing (they stage and run cells)
  constexpr int H = NKB * 32, GH = G * H, SLOTS = H / 8;      // 16-byte slots per row of an image
  constexpr int IMG = WR_ROWS * SLOTS;                        // slots of one image (hi or lo) of a 16-row tile
  constexpr int SEGS = IMG / 64;                              // DMA pieces (64 slots = 1 KB) per image
  constexpr int PW = 2 * SEGS / 8;                            // ... per wave and tile: 4 (H 512), 2 (H 256)
  constexpr int XSZ = 4 * WR_ROWS * 32;
  extern __shared__ __attribute__((aligned(16))) float rl_smem[];
  bf16x8* sm = reinterpret_cast<bf16x8*>(rl_smem);            // [buf][hi | lo][IMG]
  float* X = rl_smem + 4 * IMG * 4;                           // [buf][gate][row][unit]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);    // the same number, provably wave-uniform (scalar address math)
  const int l31 = lane & 31, lhi = lane >> 5, l15 = lane & 15, kq = lane >> 4;
  const int gate = wave & 3, uh = wave >> 2;
  constexpr int ncg = H / 32;
  const int nrg = w.nrg;
  const int b = blockIdx.x, idx = b >> 3;
  const int cg = idx % ncg, rg = (b & 7) + 8 * (idx / ncg);
  const int u0 = cg * 32, u = u0 + l31;
  const int ntiles = (p.live + WR_ROWS - 1) / WR_ROWS;
  if (rg >= ntiles) return;
#if WR_EXP_STAMPS
  const unsigned long long t00_ = __builtin_amdgcn_s_memtime();
#endif
  const int nt = (ntiles - rg + nrg - 1) / nrg;
  const int last_row = p.live - 1;

  // DMA piece j of this wave: 64 consecutive slots of one image of the tile; this lane's slot q = (row r, position ps)
  // receives logical slot ps ^ r of that row.  Rows past the live prefix are read as they are (the images are padded
  // to whole tiles; a row's garbage stays in that row of the product, which is never stored).
  unsigned dma_voff[PW];                                      // byte offset of the lane's 16 bytes from the tile's first row
#pragma unroll
  for (int j = 0; j < PW; ++j) {
    const int pc = wave * PW + j, im = pc / SEGS, q = (pc % SEGS) * 64 + lane;
    const int r = q / SLOTS, ps = q % SLOTS;
    dma_voff[j] = (unsigned)(im * w.img_lo * 2 + r * (H * 2) + 16 * (ps ^ r));
  }
  auto stage_dma = [&](int j, int n, int buf) {               // buf: compile-time or scalar
#if !WR_EXP_NOSTAGELOAD
    const int row0 = (rg + nrg * min(n, nt - 1)) * WR_ROWS;   // scalar
    const char* base = reinterpret_cast<const char*>(w.img_in) + (size_t)row0 * (H * 2);
    const int pc = wave_u * PW + j;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + dma_voff[j]),
                                     (__attribute__((address_space(3))) void*)(sm + buf * 2 * IMG + (pc / SEGS) * IMG + (pc % SEGS) * 64),
                                     16, 0, 0);
#endif
  };
#pragma unroll
  for (int j = 0; j < PW; ++j) stage_dma(j, 0, 0);

  // this wave's weights: gate `gate`, units u0 + 16 uh + (0..15), as B fragments (lane: unit l15, k = 32 kb + 8 kq + 0..7),
  // from the image rnn_large_wimg_kernel wrote
  i32x4 bh[NKB], bl[NKB];
  {
    const i32x4* src = w.wimg + (size_t)(cg * 8 + wave_u) * NKB * 2 * 64 + lane;     // (GRU's gate-3 waves: an unwritten slice of the image, never multiplied)
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
#if WR_EXP_NOB
      bh[kb] = i32x4{kb, lane, 3, 4}; bl[kb] = i32x4{kb, 1, lane, 4}; (void)src;
#else
      bh[kb] = src[(kb * 2 + 0) * 64];
      bl[kb] = src[(kb * 2 + 1) * 64];
#endif
    }
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) asm volatile("" : "+a"(bh[kb]), "+a"(bl[kb]));       // from here on accumulator-file tuples
  }
  float bq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int g = 0; g < G; ++g) bq[g] = p.b_hh[g * H + u];

  // The cell's row of a tile: m0 + 2 wave + lhi, unit u.  Its operands are addressed as (scalar base) + (32-bit lane
  // offset): one multiply-add per array instead of 64-bit arithmetic per load -- the kernel's vector issue is as busy
  // as its matrix pipe.  xadd and c of tile n+1 are requested during tile n.
  const int crow = wave * 2 + lhi;
  const float inv_xdiv = 1.f / (float)p.xdiv;
  const char* xadd_b = reinterpret_cast<const char*>(p.xadd);
  const char* xproj_b = reinterpret_cast<const char*>(p.xproj);
  const char* cprev_b = reinterpret_cast<const char*>(CELL == CELL_LSTM ? p.c_prev : p.h_prev);   // the cell's state operand: c (LSTM), h (GRU)
  auto ldf = [](const char* base, unsigned off) { return *reinterpret_cast<const float*>(base + off); };
  float xa[4], cp;
  auto cell_prefetch = [&](int n) {
    const int mc = min((rg + nrg * min(n, nt - 1)) * WR_ROWS + crow, last_row);
#if WR_EXP_NOEPILOAD
#pragma unroll
    for (int g = 0; g < 4; ++g) xa[g] = 0.1f * g + 1e-6f * mc;
    cp = 0.25f;
#else
#pragma unroll
    for (int g = 0; g < 4; ++g) xa[g] = 0.f;
    if constexpr (XADD) {
      const unsigned o = (unsigned)(mc * GH + u) * 4u;
#pragma unroll
      for (int g = 0; g < G; ++g) xa[g] = ldf(xadd_b, o + g * (H * 4));
    } else {
#pragma unroll
      for (int g = 0; g < 4; ++g) xa[g] = 0.f;
    }
    cp = ldf(cprev_b, (unsigned)(mc * H + u) * 4u);
#endif
  };
  cell_prefetch(0);
  __syncthreads();                                            // tile 0's image has landed (the barrier drains the DMA)

  // fragment reads: slot (4 kb + kq) ^ l15 of row l15 = 16 (kb >> 2) + 4 ((kb & 3) ^ (l15 >> 2)) + (kq ^ (l15 & 3)):
  // four lane addresses (kb & 3), everything else is the instruction's immediate offset
  const bf16x8* fr[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) fr[q] = sm + l15 * SLOTS + 4 * (q ^ (l15 >> 2)) + (kq ^ (l15 & 3));
  float* const xw = X + (gate * WR_ROWS + 4 * kq) * 32 + 16 * uh + l15;      // + r * 32: this lane's accumulator rows
  const float* const xr = X + crow * 32 + l31;                              // + g * WR_ROWS * 32: this thread's cell
#if WR_EXP_STAMPS
  unsigned sacc_[6] = {0, 0, 0, 0, 0, 0};
  unsigned long long tprev_ = __builtin_amdgcn_s_memtime();
  const unsigned long long tstart_ = tprev_;
#endif
  // the cell of tile n-1 runs inside the MFMA loop of tile n (one tile behind: the matrix pipe has work while the
  // activations are computed): its summed pre-activations, its c and its output offset travel in pre / cprev / oprev
  float pre[4] = {0.f, 0.f, 0.f, 0.f}, cprev = 0.f;
  unsigned oprev = 0;
  bool vprev = false;                                         // no row yet
  char* hout_b = reinterpret_cast<char*>(p.h_out);
  char* cout_b = reinterpret_cast<char*>(p.c_out);
  char* ihi_b = reinterpret_cast<char*>(w.img_out);
  char* ilo_b = reinterpret_cast<char*>(w.img_out + w.img_lo);
  auto cell_finish = [&]() {
    float hnew, cnew = 0.f;
    if constexpr (CELL == CELL_LSTM) {
      const float ig = sigmoid_fast(pre[0]);
      const float fg = sigmoid_fast(pre[1]);
      const float gg = tanh_fast(pre[2]);
      const float og = sigmoid_fast(pre[3]);
      cnew = fg * cprev + ig * gg;
      hnew = og * tanh_fast(cnew);
    } else {                                                  // pre = (r, z, x-side of n, h-side of n); cprev = h
      const float rgate = sigmoid_fast(pre[0]);
      const float zg = sigmoid_fast(pre[1]);
      const float nn = tanh_fast(pre[2] + rgate * pre[3]);
      hnew = (1.f - zg) * nn + zg * cprev;
    }
    const __bf16 hh = (__bf16)hnew;
    const __bf16 hl = (__bf16)(hnew - (float)hh);
    if (vprev) {
      *reinterpret_cast<float*>(hout_b + oprev) = hnew;
      if constexpr (CELL == CELL_LSTM) *reinterpret_cast<float*>(cout_b + oprev) = cnew;
      *reinterpret_cast<__bf16*>(ihi_b + (oprev >> 1)) = hh;
      *reinterpret_cast<__bf16*>(ilo_b + (oprev >> 1)) = hl;
    }
  };
  for (int n = 0; n < nt; ++n) {
    WR_STAMP(4);
    const int m = (rg + nrg * n) * WR_ROWS + crow;
    const int mc = min(m, last_row);
    float x[4], cc = cp;
#pragma unroll
    for (int g = 0; g < 4; ++g) x[g] = xa[g];
    // these copies are the last use of an ordinary load's result before the barrier: while an LDS-DMA is in flight
    // hipcc drains the whole queue (vmcnt(0)) at any such use, so the DMA is issued after them
    asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(cc));
    WR_STAMP(5);
    float xq[4] = {0.f, 0.f, 0.f, 0.f};
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
    const int boff = (n & 1) * 2 * IMG;                        // scalar: this tile's buffer
    // A fragments WR_PF K blocks ahead of their MFMAs (a ring of WR_PF + 1 register sets)
    i32x4 fh[WR_PF + 1], fl[WR_PF + 1];
#pragma unroll
    for (int d = 0; d < WR_PF; ++d) {
      fh[d] = __builtin_bit_cast(i32x4, fr[d & 3][boff + 16 * (d >> 2)]);
      fl[d] = __builtin_bit_cast(i32x4, fr[d & 3][boff + IMG + 16 * (d >> 2)]);
    }
    WR_STAMP(0);
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
      // memory instructions of the tile, spread over the MFMA loop: the other buffer's DMA (its readers passed barrier
      // n-1), the sentence-shared x-side pre-activations (L2-resident), the next tile's per-row operands, the cell of
      // the previous tile
      if (kb < PW) stage_dma(kb, n + 1, (n + 1) & 1);
#if !WR_EXP_NOEPILOAD
      if (kb == PW) {
        int q = (int)((float)mc * inv_xdiv);                  // mc / xdiv (< xmod: checked by the launcher): float estimate, then exact
        const int rem = mc - q * p.xdiv;
        q += rem >= p.xdiv ? 1 : (rem < 0 ? -1 : 0);
        const unsigned o = (unsigned)(q * GH + u) * 4u;
#pragma unroll
        for (int g = 0; g < G; ++g) xq[g] = ldf(xproj_b, o + g * (H * 4));
      }
#endif
      if (kb == PW + 1) cell_prefetch(n + 1);
#if !WR_EXP_NOCELL
      if (kb == PW + 3) cell_finish();                        // tile n-1
#endif
      if (kb + WR_PF < NKB) {
        const int k2 = kb + WR_PF;
        fh[k2 % (WR_PF + 1)] = __builtin_bit_cast(i32x4, fr[k2 & 3][boff + 16 * (k2 >> 2)]);
        fl[k2 % (WR_PF + 1)] = __builtin_bit_cast(i32x4, fr[k2 & 3][boff + IMG + 16 * (k2 >> 2)]);
      }
      const i32x4 ah = fh[kb % (WR_PF + 1)], al = fl[kb % (WR_PF + 1)];
      // s_nop 1: a just-written VGPR operand -> MFMA
#if WR_EXP_NOMFMA
      acc0[kb & 3] += __builtin_bit_cast(f32x4, ah)[0] + __builtin_bit_cast(f32x4, al)[1];
      if (false)
#endif
      if (G == 4 || (wave_u & 3) < G)                         // wave-uniform: GRU's gate-3 waves multiply nothing
      asm volatile("s_nop 1\n\t"
                   "v_mfma_f32_16x16x32_bf16 %0, %3, %6, %0\n\t"
                   "v_mfma_f32_16x16x32_bf16 %1, %4, %5, %1\n\t"
                   "v_mfma_f32_16x16x32_bf16 %2, %3, %5, %2"
                   : "+v"(acc0), "+v"(acc1), "+v"(acc2) : "v"(ah), "v"(al), "a"(bh[kb]), "a"(bl[kb]));
    }
    asm volatile("s_nop 15" : "+v"(acc0), "+v"(acc1), "+v"(acc2));      // MFMA result -> any other reader: wait states
    WR_STAMP(1);
    const f32x4 acc = (acc0 + acc1) + acc2;
#if WR_EXP_NOCELL
    if (m < p.live) p.h_out[(long)m * H + u] = acc[0] + acc[3] + x[0] + x[1] + x[2] + x[3] + cc + xq[0] + xq[3];
    __syncthreads();
    continue;
#endif
    // acc[r] = gate `gate` of row 4 kq + r, unit 16 uh + l15
    const int xoff = (n & 1) * XSZ;
    if (G == 4 || gate < G) {
#pragma unroll
      for (int r = 0; r < 4; ++r) xw[xoff + r * 32] = acc[r];
    }
    // one barrier per tile: X(n) complete; the other buffer's image landed (hipcc drains an LDS-DMA in flight at a
    // barrier); this buffer's reads over.  X is double-buffered: X(n+2) is written after barrier n+1, which every
    // wave passes only after it has read X(n).
    __syncthreads();
    WR_STAMP(2);
    if constexpr (CELL == CELL_LSTM) {
#pragma unroll
      for (int g = 0; g < 4; ++g) pre[g] = xr[xoff + g * WR_ROWS * 32] + ((x[g] + xq[g]) + bq[g]);
    } else {
#pragma unroll
      for (int g = 0; g < 2; ++g) pre[g] = xr[xoff + g * WR_ROWS * 32] + ((x[g] + xq[g]) + bq[g]);
      pre[2] = x[2] + xq[2];
      pre[3] = xr[xoff + 2 * WR_ROWS * 32] + bq[2];
    }
    cprev = cc;
    oprev = (unsigned)(mc * H + u) * 4u;
    vprev = m < p.live;
    WR_STAMP(3);
  }
#if !WR_EXP_NOCELL
  cell_finish();
#endif
#if WR_EXP_STAMPS
  if ((blockIdx.x == 0 || blockIdx.x == 100) && (tid == 0 || tid == 448))
    printf("wreg stamps block %d tid %d: tiles %d, prologue %u, per tile: copies %u, top %u, kloop %u, barrier %u, cell %u, back-edge %u (s_memtime ticks)\n",
           (int)blockIdx.x, tid, nt, (unsigned)(tstart_ - t00_), sacc_[5] / nt, sacc_[0] / nt, sacc_[1] / nt, sacc_[2] / nt, sacc_[3] / nt, sacc_[4] / nt);
#endif
}

// rows below which the 128-row-tile kernel is the faster one (too few 16-row tiles per row group); env
// CVAE_RNN_LARGE_WREG_MIN, CVAE_RNN_LARGE_WREG=0 turns the kernel off; cvae_rnn_large_wreg() is the runtime form
static int g_wreg_min = -1;
static int wreg_min_rows() {
  if (g_wreg_min < 0) {
    const char* e = getenv("CVAE_RNN_LARGE_WREG_MIN");
    g_wreg_min = e ? atoi(e) : 64;
    const char* on = getenv("CVAE_RNN_LARGE_WREG");
    if (on && on[0] == '0') g_wreg_min = 1 << 30;
  }
  return g_wreg_min;
}
int rnn_large_wreg_set(int min_rows) {
  const int prev = wreg_min_rows();
  if (min_rows >= 0) g_wreg_min = min_rows;
  return prev;
}

// (xmod, xdiv): the kernel takes row r's x-side row as r / xdiv, so every live row must lie below xdiv * xmod; its
// 32-bit byte offsets need live * 4H * 4 < 2^32
bool rnn_large_wreg_supported(int cell, int H, int live, int xmod, int xdiv, const float* h, const float* W_hh) {
  return split_bf16_enabled() && (cell == CELL_LSTM || cell == CELL_GRU) && (H == 512 || H == 256) && live >= wreg_min_rows() &&
         (long)live <= (long)xmod * xdiv && (long)live * 16 * H < (1L << 32) &&
         aligned16(h) && aligned16(W_hh) && persistent_grid_blocks() / 2 / (H / 32) >= 8;
}

// an image pair of R rows: each half padded to whole 16-row tiles (the kernel's DMA reads whole tiles)
long rnn_large_image_lo(int R, int H) { return (long)((R + WR_ROWS - 1) / WR_ROWS * WR_ROWS) * H; }
size_t rnn_large_image_bytes(int R, int H) { return (size_t)2 * rnn_large_image_lo(R, H) * sizeof(__bf16); }

// (hi, lo) image of rows [0, rows) of h ([rows, H] fp32); img_lo = elements between the two halves of the image
int launch_rnn_large_split(const float* h, int rows, int H, void* img, long img_lo, hipStream_t s) {
  CVAE_REQUIRE(h && img && rows > 0 && H % 8 == 0 && aligned16(h) && aligned16(img) && img_lo % 8 == 0, "rnn_large_split: bad argument");
  const long n8 = (long)rows * H / 8;
  hipLaunchKernelGGL(rnn_large_split_kernel, dim3((unsigned)cdiv(n8, 256)), dim3(256), 0, s, n8, h, static_cast<__bf16*>(img), img_lo);
  CVAE_LAUNCH_CHECK("rnn_large_split_kernel");
  return 0;
}

size_t rnn_large_wimg_bytes(int H) { return (size_t)4 * H * H * 4; }      // 4H x H (hi, lo) bf16

// W_hh ([4H, H] fp32) -> the step kernel's fragment image
int launch_rnn_large_wimg(int cell, const float* W_hh, int H, void* wimg, hipStream_t s) {
  CVAE_REQUIRE(W_hh && wimg && (H == 512 || H == 256) && aligned16(W_hh) && aligned16(wimg), "rnn_large_wimg: bad argument");
  hipLaunchKernelGGL(rnn_large_wimg_kernel, dim3(H / 32), dim3(WR_THREADS), 0, s, H, cell == CELL_GRU ? 3 : 4, W_hh,
                     static_cast<i32x4*>(wimg));
  CVAE_LAUNCH_CHECK("rnn_large_wimg_kernel");
  return 0;
}

// one step of rows [0, live) on the weights-in-registers kernel; img_in = split image of h_prev, img_out receives h_out's;
// wimg = launch_rnn_large_wimg's image of W_hh
int launch_rnn_large_wreg_step(int cell, int live, int H, const float* h_prev, const float* c_prev, const void* wimg,
                               const float* b_hh, const float* xproj, int xmod, int xdiv, const float* xadd, float* h_out,
                               float* c_out, const void* img_in, void* img_out, long img_lo, hipStream_t s) {
  if (live <= 0) return 0;
  CVAE_REQUIRE(h_prev && wimg && b_hh && xproj && h_out && img_in && img_out && xmod > 0 && xdiv > 0,
               "rnn_large_wreg_step: null operand");
  CVAE_REQUIRE(cell == CELL_GRU || (cell == CELL_LSTM && c_prev && c_out), "rnn_large_wreg_step: bad cell, or LSTM without c");
  CVAE_REQUIRE(H == 512 || H == 256, "rnn_large_wreg_step: H=%d (256 or 512)", H);
  CVAE_REQUIRE(aligned16(img_in) && aligned16(img_out) && img_lo % 8 == 0 && img_lo >= rnn_large_image_lo(live, H),
               "rnn_large_wreg_step: image misaligned or too small");
  CVAE_REQUIRE((long)live <= (long)xmod * xdiv && (long)live * 16 * H < (1L << 32), "rnn_large_wreg_step: rows beyond xdiv * xmod or 32-bit offsets");
  const int ncg = H / 32, nrg = persistent_grid_blocks() / 2 / ncg / 8 * 8;     // row groups: one block per CU
  CVAE_REQUIRE(nrg >= 8, "rnn_large_wreg_step: %d CUs are too few", persistent_grid_blocks() / 2);
  RnnWregArgs a = {{h_prev, c_prev, nullptr, b_hh, xproj, xadd, h_out, c_out, live, H, xmod, xdiv},
                   static_cast<const __bf16*>(img_in), static_cast<__bf16*>(img_out), img_lo, nrg,
                   static_cast<const i32x4*>(wimg)};
  ProfScope prof(PK_RNN_LARGE, s, 2.0 * live * (cell == CELL_GRU ? 3 : 4) * H * (double)H);
  const dim3 gridw(ncg * nrg);
  const size_t ldsw = (size_t)4 * WR_ROWS * (H / 8) * 16 + 2 * 4 * WR_ROWS * 32 * sizeof(float);     // 80 KB (H 512) / 48 KB
  static bool attrw = false;
  if (!attrw) {
    const void* fs[8] = {reinterpret_cast<const void*>(&rnn_large_wreg_kernel<16, true, CELL_LSTM>),
                         reinterpret_cast<const void*>(&rnn_large_wreg_kernel<16, false, CELL_LSTM>),
                         reinterpret_cast<const void*>(&rnn_large_wreg_kernel<8, true, CELL_LSTM>),
                         reinterpret_cast<const void*>(&rnn_large_wreg_kernel<8, false, CELL_LSTM>),
                         reinterpret_cast<const void*>(&rnn_large_wreg_kernel<16, true, CELL_GRU>),
                         reinterpret_cast<const void*>(&rnn_large_wreg_kernel<16, false, CELL_GRU>),
                         reinterpret_cast<const void*>(&rnn_large_wreg_kernel<8, true, CELL_GRU>),
                         reinterpret_cast<const void*>(&rnn_large_wreg_kernel<8, false, CELL_GRU>)};
    for (const void* f : fs)
      CVAE_TRY(check_hip(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024), "rnn_large wreg LDS attribute"));
    attrw = true;
  }
  const dim3 blockw(WR_THREADS);
#define WR_LAUNCH(CELL_)                                                                                         \
  do {                                                                                                           \
    if (H == 512 && xadd) hipLaunchKernelGGL((rnn_large_wreg_kernel<16, true, CELL_>), gridw, blockw, ldsw, s, a);   \
    else if (H == 512) hipLaunchKernelGGL((rnn_large_wreg_kernel<16, false, CELL_>), gridw, blockw, ldsw, s, a);    \
    else if (xadd) hipLaunchKernelGGL((rnn_large_wreg_kernel<8, true, CELL_>), gridw, blockw, ldsw, s, a);          \
    else hipLaunchKernelGGL((rnn_large_wreg_kernel<8, false, CELL_>), gridw, blockw, ldsw, s, a);                   \
  } while (0)
  if (cell == CELL_GRU) WR_LAUNCH(CELL_GRU);
  else WR_LAUNCH(CELL_LSTM);
#undef WR_LAUNCH
  CVAE_LAUNCH_CHECK("rnn_large_wreg_kernel");
  return 0;
}

bool rnn_large_fused_supported(int cell, int H, const float* h, const float* W_hh) {
  static int on = -1;
  if (on < 0) {
    const char* e = getenv("CVAE_FUSED_LARGE_STEP");
    on = (e && e[0] == '0') ? 0 : 1;
  }
  return on && H % RL_BK == 0 && H % RL_BU == 0 && aligned16(h) && aligned16(W_hh);
}

// rows [0, live) of one step; h_prev / h_out (and c) are [R, H] slabs
int launch_rnn_large_step(int cell, int live, int H, const float* h_prev, const float* c_prev, const float* W_hh,
                          const float* b_hh, const float* xproj, int xmod, int xdiv, const float* xadd, float* h_out,
                          float* c_out, hipStream_t s) {
  if (live <= 0) return 0;
  CVAE_REQUIRE(h_prev && W_hh && b_hh && xproj && h_out && xmod > 0 && xdiv > 0, "rnn_large_step: null operand");
  CVAE_REQUIRE(cell == CELL_GRU || (c_prev && c_out), "rnn_large_step: LSTM needs c");
  CVAE_REQUIRE(H % RL_BK == 0 && H % RL_BU == 0, "rnn_large_step: H=%d must be a multiple of 32", H);
  RnnLargeArgs a = {h_prev, c_prev, W_hh, b_hh, xproj, xadd, h_out, c_out, live, H, xmod, xdiv};
  const int G = cell == CELL_GRU ? 3 : 4;
  const dim3 grid((H / RL_BU) * ((cdiv(live, RL_BM) + 7) / 8 * 8)), block(256);
  const size_t lds = (size_t)2 * (RL_BM + G * RL_BU) * RL_LD * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    CVAE_TRY(check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(&rnn_large_step_kernel<CELL_GRU>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024), "rnn_large LDS attribute"));
    CVAE_TRY(check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(&rnn_large_step_kernel<CELL_LSTM>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024), "rnn_large LDS attribute"));
    attr_set = true;
  }
  ProfScope prof(PK_RNN_LARGE, s, 2.0 * live * G * H * (double)H);
  if (split_bf16_enabled() && H % RB_BK == 0) {
    const size_t lds3 = (size_t)2 * (2 * RL_BM * 4 + 2 * G * RL_BU * 4) * 16;      // 64 KB (LSTM) / 56 KB (GRU)
    static bool attr3 = false;
    if (!attr3) {
      CVAE_TRY(check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(&rnn_large_step_bf3_kernel<CELL_GRU>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024), "rnn_large bf3 LDS attribute"));
      CVAE_TRY(check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(&rnn_large_step_bf3_kernel<CELL_LSTM>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024), "rnn_large bf3 LDS attribute"));
      attr3 = true;
    }
    if (cell == CELL_GRU) hipLaunchKernelGGL(rnn_large_step_bf3_kernel<CELL_GRU>, grid, block, lds3, s, a);
    else hipLaunchKernelGGL(rnn_large_step_bf3_kernel<CELL_LSTM>, grid, block, lds3, s, a);
    CVAE_LAUNCH_CHECK("rnn_large_step_bf3_kernel");
    return 0;
  }
  if (cell == CELL_GRU) hipLaunchKernelGGL(rnn_large_step_kernel<CELL_GRU>, grid, block, lds, s, a);
  else hipLaunchKernelGGL(rnn_large_step_kernel<CELL_LSTM>, grid, block, lds, s, a);
  CVAE_LAUNCH_CHECK("rnn_large_step_kernel");
  return 0;
}

}  // namespace cvae
